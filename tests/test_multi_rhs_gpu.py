"""post-solve reuse of the factor with BLOCKS of right-hand sides -- dlg_solve_multi
(16 right-hand sides per pass over the resident factor, matrix cores for the off-diagonal products)
and dlg_pseudoinverse_chunk = inv(JtJ) Jt[:, rows] (reference pseudoinverse_J_dense / _sparse,
dogleg.c:1831-1921; cholmod_spsolve, dogleg.c:2864-2868).  Checked against the ORACLE's solves
(orc_sparse_solve, orc_dpptrs_L), column by column."""
import numpy as np
import pytest

from libdogleg_amd import capi
from libdogleg_amd.ctypes_defs import dptr, iptr
from tests import oracle_api as oa

pytestmark = pytest.mark.gpu


def _oracle_sparse_solves(prob, Jx, lam, rhs):
    O = oa.oracle()
    Jp, Ji = prob.pattern()
    F = O.orc_sparse_analyze(prob.N, prob.M, iptr(Jp), iptr(Ji))
    assert O.orc_sparse_factorize(F, iptr(Jp), iptr(Ji), dptr(Jx), lam) == prob.N
    out = np.zeros_like(rhs)
    for k in range(rhs.shape[0]):
        O.orc_sparse_solve(F, dptr(np.ascontiguousarray(rhs[k])), dptr(out[k]))
    O.orc_sparse_free(F)
    return out


@pytest.mark.parametrize("shape,lam", [((5, 40, 300), 1e-3), ((49, 900, 10000), 0.0), ((199, 3600, 40000), 0.0)],
                         ids=["tiny", "medium", "large"])
def test_sparse_blocked_solve_matches_oracle(gpu, shape, lam):
    rng = np.random.default_rng(3)
    prob = oa.BAProblem(*shape, seed=11)
    p = prob.p0()
    x, Jx = prob.eval(p)
    Jp, Ji = prob.pattern()
    be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
    be.set_pattern(Jp, Ji)
    be.set_p(0, p)
    be.upload(0, x, Jx)
    be.eval(0)
    assert be.factorize(0, lam)
    nrhs = 37                                    # two full blocks of 16 and a ragged one
    rhs = rng.standard_normal((nrhs, prob.N))
    u = be.solve_multi(0, rhs)
    ref = _oracle_sparse_solves(prob, Jx, lam, rhs)
    err = np.max(np.linalg.norm(u - ref, axis=1) / np.linalg.norm(ref, axis=1))
    print(f"{shape}: blocked solve of {nrhs} right-hand sides, worst relative |u - oracle| = {err:.2e}")
    assert err <= 1e-10
    # the one-at-a-time path solves the same systems
    u1 = be.solve_with_factor(0, rhs[:3])
    assert np.max(np.abs(u1 - u[:3])) <= 1e-10 * np.max(np.abs(u[:3]))
    # empty and single right-hand side
    assert be.solve_multi(0, rhs[:0]).shape == (0, prob.N)
    assert np.max(np.abs(be.solve_multi(0, rhs[:1]) - u[:1])) == 0.0
    be.close()


def test_sparse_pseudoinverse_chunk_matches_oracle(gpu):
    """inv(JtJ) Jt[:, i0:i0+n]: the chunk the reference's outlier / confidence code asks for"""
    prob = oa.BAProblem(49, 900, 10000, seed=5)
    p = prob.p0()
    x, Jx = prob.eval(p)
    Jp, Ji = prob.pattern()
    be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
    be.set_pattern(Jp, Ji)
    be.set_p(0, p)
    be.upload(0, x, Jx)
    be.eval(0)
    lam, n2g = be.gauss_newton(0, 0.0)            # the factor a finished solve leaves behind
    i0, n = 4321, 41
    pinv = be.pseudoinverse_chunk(0, i0, i0 + n)
    rhs = np.zeros((n, prob.N))
    for k in range(n):
        r = i0 + k
        rhs[k, Ji[Jp[r]:Jp[r+1]]] = Jx[Jp[r]:Jp[r+1]]
    ref = _oracle_sparse_solves(prob, Jx, lam, rhs)
    err = np.max(np.linalg.norm(pinv - ref, axis=1) / np.linalg.norm(ref, axis=1))
    print(f"pseudo-inverse chunk of {n} measurements: worst relative difference to the oracle {err:.2e}")
    assert err <= 1e-10
    # size-independent property: J pinv(J)[:, rows] restricted to the rows is symmetric (a projector block)
    Jd = np.zeros((n, prob.N))
    Jd[:] = rhs
    Pblk = Jd @ pinv.T
    assert np.max(np.abs(Pblk - Pblk.T)) <= 1e-10
    with pytest.raises(capi.DlgError):
        be.pseudoinverse_chunk(0, 0, prob.M + 1)
    be.close()


@pytest.mark.parametrize("M,N", [(400, 50), (3000, 257), (5000, 1030)])
def test_dense_blocked_solve_and_pseudoinverse_match_oracle(gpu, M, N):
    O = oa.oracle()
    rng = np.random.default_rng(7)
    dp = oa.DenseProblem(M=M, N=N, seed=11)
    p = dp.p0()
    x, J = dp.eval(p)
    be = capi.Backend(capi.DLG_DENSE, N, M)
    be.set_p(0, p)
    be.upload(0, x, J)
    be.eval(0)
    lam = 1e-3
    assert be.factorize(0, lam)
    # the oracle's packed factor of the same matrix (dpptrf restated; JtJ by BLAS: the rank-1 loop is checked elsewhere)
    A = J.T @ J + lam * np.eye(N)
    ap = np.ascontiguousarray(A[np.triu_indices(N)])
    assert O.orc_dpptrf_L(N, dptr(ap)) == 0
    nrhs = 21
    rhs = rng.standard_normal((nrhs, N))
    u = be.solve_multi(0, rhs)
    ref = rhs.copy()
    for k in range(nrhs):
        O.orc_dpptrs_L(N, dptr(ap), dptr(ref[k]))
    err = np.max(np.linalg.norm(u - ref, axis=1) / np.linalg.norm(ref, axis=1))
    assert err <= 1e-10, err
    i0, n = 17, 19
    pinv = be.pseudoinverse_chunk(0, i0, i0 + n)
    refp = np.ascontiguousarray(J[i0:i0 + n]).copy()
    for k in range(n):
        O.orc_dpptrs_L(N, dptr(ap), dptr(refp[k]))
    errp = np.max(np.linalg.norm(pinv - refp, axis=1) / np.linalg.norm(refp, axis=1))
    print(f"dense {M}x{N}: blocked solve {err:.2e}, pseudo-inverse chunk {errp:.2e} (relative, vs orc_dpptrs_L)")
    assert errp <= 1e-10
    be.close()


def test_blocked_solve_state_errors(gpu):
    prob = oa.BAProblem(5, 40, 300, seed=2)
    be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
    be.set_pattern(*prob.pattern())
    with pytest.raises(capi.DlgError):                         # no factor yet
        be.solve_multi(0, np.zeros((2, prob.N)))
    be.close()
