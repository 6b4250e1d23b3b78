"""CPU tests: the oracle against the reference's known answers and against numpy.
No GPU needed."""
import ctypes as C
import json
import math
import os
import numpy as np
import pytest

from libdogleg_amd.ctypes_defs import dptr, iptr, STEP_NAMES
from tests import oracle_api as oa

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _sample_setup(kind):
    P = oa.problems()
    p0 = np.zeros(6)
    P.sample_init(dptr(p0))
    prm = oa.default_params()
    prm.max_iterations = 8                                   # sample.c:365
    cookie = None
    k = kind
    cb = {"sparse": "sample_cb_sparse", "dense": "sample_cb_dense"}.get(kind, "sample_cb_products")
    if kind.startswith("products"):
        k = "products"
        if kind == "products_packed_upper":
            prm.JtJ_packed = True
            prm.JtJ_upper = True
        cookie = C.cast(C.pointer(prm), C.c_void_p)
    return P, p0, prm, cookie, oa.fn_addr(P, cb), k


def test_sample_fixture_is_the_glibc_stream():
    """the committed measurement fixture == srandom(0)/random() stream == SURVEY known answers"""
    g = json.load(open(os.path.join(GOLD, "sample_measurements.json")))
    t = json.load(open(os.path.join(GOLD, "sample_trace.json")))
    P = oa.problems()
    p0 = np.zeros(6)
    P.sample_init(dptr(p0))
    m = np.zeros(100)
    P.sample_get_measurements(dptr(m))
    assert [float(v).hex() for v in m] == g["measurements_hex"]
    assert [float(v).hex() for v in p0] == g["p0_hex"]
    assert np.allclose(m[:3], t["measurements_first3"], rtol=0, atol=1e-12)
    assert np.allclose(p0, t["p0"], rtol=0, atol=1e-16)


@pytest.mark.parametrize("kind", ["sparse", "dense", "products_packed_upper", "products_unpacked"])
def test_reference_check_assertions(kind):
    """what `sample --check <mode>` asserts (reference check.sh:11-14, sample.c:424-458)"""
    P, p0, prm, cookie, cb, k = _sample_setup(kind)
    r, p, tr = oa.oracle_solve(k, p0, 6, 100, 600 if kind == "sparse" else 0, cb, cookie, prm)
    assert r >= 0
    assert np.all(np.abs(p - np.arange(1, 7)) < 5e-2)


@pytest.mark.parametrize("kind", ["sparse", "dense", "products_packed_upper", "products_unpacked"])
def test_oracle_reproduces_reference_trace(kind):
    """every vnlog field at %g precision, every point handed to the
    callback, the final p"""
    t = json.load(open(os.path.join(GOLD, "sample_trace.json")))
    P, p0, prm, cookie, cb, k = _sample_setup(kind)
    r, p, tr = oa.oracle_solve(k, p0, 6, 100, 600 if kind == "sparse" else 0, cb, cookie, prm)
    assert tr.ncallbacks == t["ncallbacks"]
    assert tr.ntrials == len(t["vnlog"])
    assert abs(r - t["norm2x_final"]) < 5e-7
    assert np.max(np.abs(p - np.array(t["p_final"]))) < 1e-11
    ev = np.array(t["eval_points"])
    # eval 0 is the start point; eval i (i>=1) is the trial point of trial i-1.  The survey lists
    # 8 evaluations; the 8th trial (terminal, un-applied) is never evaluated.
    assert np.max(np.abs(p0 - ev[0])) < 1e-15
    for i in range(1, len(ev)):
        assert np.max(np.abs(tr.p_trial[i-1] - ev[i])) < 2e-11, i

    def g6(v):
        return float("%g" % v)
    for rec, row in zip(tr.trials(), t["vnlog"]):
        (it, acc, n2b, n2a, lc, lgn, li, kk, sl, stype, _dir, ei, oi, rho, trb, tra) = row
        assert rec["iteration"] == it
        assert (1 if rec["accepted"] else 0) == acc
        assert STEP_NAMES[rec["step_type"]] == stype
        assert g6(rec["norm2x_before"]) == n2b
        if n2a is not None:
            assert g6(rec["norm2x_after"]) == n2a
        assert abs(math.sqrt(rec["norm2_cauchy"]) - lc) <= 2e-5 * lc + 1e-12
        if lgn is not None:
            assert abs(math.sqrt(rec["norm2_gn"]) - lgn) <= 2e-5 * lgn
        if kk is not None:
            assert g6(rec["k_cauchy_to_gn"]) == kk
            assert abs(math.sqrt(rec["norm2_step"]) - li) <= 2e-5 * li
        assert abs(math.sqrt(rec["norm2_step"]) - sl) <= 2e-5 * sl
        if rec["accepted"] != 2:
            assert g6(rec["expected_improvement"]) == ei
        else:
            # the terminal record carries the computed value too (dogleg.c:1267-1269 precedes the -1 of
            # 1289-1296); at convergence it is a difference of nearly equal tiny terms: 4 digits
            assert abs(rec["expected_improvement"] - ei) <= 1e-4 * abs(ei)
        if rec["accepted"] != 2:
            assert g6(rec["observed_improvement"]) == oi
            assert g6(rec["rho"]) == rho
            assert g6(rec["trustregion_after"]) == tra
        assert g6(rec["trustregion_before"]) == trb


def test_primitives_against_numpy():
    O = oa.oracle()
    rng = np.random.default_rng(0)
    M, N = 57, 13
    J = rng.standard_normal((M, N))
    x = rng.standard_normal(M)
    v = rng.standard_normal(N)
    out = np.zeros(N)
    O.orc_dense_Jt_x(dptr(out), dptr(J), dptr(x), M, N)
    assert np.allclose(out, J.T @ x, rtol=1e-13)
    assert abs(O.orc_dense_norm2_J_v(dptr(J), dptr(v), M, N) - np.sum((J @ v) ** 2)) < 1e-10
    A = J.T @ J
    assert abs(O.orc_xt_A_x(dptr(v), dptr(np.ascontiguousarray(A)), N) - v @ A @ v) < 1e-9
    pu = np.ascontiguousarray(A[np.triu_indices(N)])
    assert abs(O.orc_xt_Apacked_upper_x(dptr(v), dptr(pu), N) - v @ A @ v) < 1e-9
    # packed rank-1 JtJ + dpptrf/dpptrs vs numpy
    ap = np.zeros(N * (N + 1) // 2)
    O.orc_dense_JtJ_packed_upper(dptr(ap), dptr(J), M, N)
    assert np.allclose(ap, pu, rtol=1e-12)
    assert O.orc_dpptrf_L(N, dptr(ap)) == 0
    Lref = np.linalg.cholesky(A)
    Lg = np.zeros((N, N))
    Lg[np.triu_indices(N)[::-1]] = ap          # row-major upper == column-major lower
    assert np.allclose(Lg, Lref, rtol=1e-11, atol=1e-12)
    b = rng.standard_normal(N)
    sol = b.copy()
    O.orc_dpptrs_L(N, dptr(ap), dptr(sol))
    assert np.allclose(sol, np.linalg.solve(A, b), rtol=1e-9)
    full = np.ascontiguousarray(A.copy())
    assert O.orc_dpotrf_L(N, dptr(full), N) == 0
    sol2 = b.copy()
    O.orc_dpotrs_L(N, dptr(full), N, dptr(sol2))
    assert np.allclose(sol2, np.linalg.solve(A, b), rtol=1e-9)
    # a non-positive-definite matrix is reported, with the LAPACK index
    bad = np.eye(4)
    bad[2, 2] = 0.0
    assert O.orc_dpotrf_L(4, dptr(np.ascontiguousarray(bad)), 4) == 3
    badp = np.ascontiguousarray(bad[np.triu_indices(4)])
    assert O.orc_dpptrf_L(4, dptr(badp)) == 3


def test_sparse_cholesky_against_dense():
    O = oa.oracle()
    prob = oa.BAProblem(5, 30, 120, seed=8)
    Jp, Ji = prob.pattern()
    x, Jx = prob.eval(prob.p0())
    N, M = prob.N, prob.M
    Jd = np.zeros((M, N))
    for r in range(M):
        Jd[r, Ji[Jp[r]:Jp[r+1]]] = Jx[Jp[r]:Jp[r+1]]
    g = np.zeros(N)
    O.orc_spmv_Jt_x(dptr(g), N, M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(x))
    assert np.allclose(g, Jd.T @ x, rtol=1e-12)
    assert abs(O.orc_norm2_J_v(M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(g)) - np.sum((Jd @ g) ** 2)) <= 1e-9 * np.sum((Jd @ g) ** 2)
    F = O.orc_sparse_analyze(N, M, iptr(Jp), iptr(Ji))
    for beta in (0.0, 1e-3):
        assert O.orc_sparse_factorize(F, iptr(Jp), iptr(Ji), dptr(Jx), beta) == N
        sol = np.zeros(N)
        O.orc_sparse_solve(F, dptr(g), dptr(sol))
        ref = np.linalg.solve(Jd.T @ Jd + beta * np.eye(N), g)
        assert np.linalg.norm(sol - ref) <= 1e-10 * np.linalg.norm(ref)
    O.orc_sparse_free(F)


def test_lambda_schedule():
    """exactly-zero columns: 0 -> 1e-10 (-> x10 ...) and sticky (dogleg.c:138,656-677,806-815)"""
    prob = oa.BAProblem(6, 40, 160, seed=7, n_zero_cols=2)
    prm = oa.default_params()
    prm.max_iterations = 6
    prm.trustregion0 = 100.0
    r, p, tr = oa.oracle_solve("sparse", prob.p0(), prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    lam = [t["lambda_"] for t in tr.trials()]
    assert r >= 0
    assert lam[0] == 1e-10 or 1e-10 in lam
    first = lam.index(1e-10)
    assert all(l >= 1e-10 for l in lam[first:])           # never decreases


def test_committed_oracle_goldens_are_current():
    """the oracle still produces the committed BA / dense traces bit for bit"""
    g = json.load(open(os.path.join(GOLD, "oracle_ba_tiny.json")))
    prob = oa.BAProblem(4, 20, 60, seed=2, eps=0.4, p0_spread=0.8)
    prm = oa.default_params()
    prm.max_iterations = 15
    prm.trustregion0 = 1.0
    r, p, tr = oa.oracle_solve("sparse", prob.p0(), prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    assert float(r).hex() == g["norm2x"]
    assert [float(v).hex() for v in p] == g["p_final_hex"]
    assert tr.ntrials == len(g["trials"])
    for i, t in enumerate(g["trials"]):
        assert [float(v).hex() for v in tr.step[i]] == t["step_hex"]
