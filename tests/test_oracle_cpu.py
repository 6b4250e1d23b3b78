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


# ---------------------------------------------------------------------------------------------
# Pinning at the 1e-10 bar with an independent leg (tests/independent.py): the REAL LAPACK
# routines the reference calls (scipy.linalg.lapack.dpptrf/dpptrs/dpotrf/dpotrs = dogleg.c:782,
# 875, 801, 889) on the dense paths, scipy's SuperLU with extended-precision refinement where the
# reference calls CHOLMOD (dogleg.c:659-664, 853).  Every trial step of an oracle trace is
# re-derived from the problem's x and J with numpy + those libraries only.
from tests import independent as ind


def _rederive_trace(tr, eval_at, sparse_pattern=None, step_tol=1e-10, dense_packed=True):
    """eval_at(p) -> (x, J) with J dense (M,N) or the CSC values of Jt.  Returns the worst
    |step_oracle - step_independent|_2 over the trials."""
    worst = 0.0
    for i, t in enumerate(tr.trials()):
        p_from = tr.p_trial[i] - tr.step[i]
        x, J = eval_at(p_from)
        if sparse_pattern is not None:
            Jp, Ji, M, N = sparse_pattern
            J = ind.csr_from_pattern(M, N, Jp, Ji, J)
        st = ind.trial_step(J, x, t["trustregion_before"], t["lambda_"], dense_packed)
        assert st["kind"] == t["step_type"], f"trial {i}: step kind {t['step_type']} vs independent {st['kind']}"
        d = float(np.linalg.norm(st["step"] - tr.step[i]))
        worst = max(worst, d)
        assert d <= step_tol, f"trial {i}: |step - independent| = {d:.3e}"
        ei = t["expected_improvement"]
        assert abs(st["expected_improvement"] - ei) <= 1e-9 * max(1.0, abs(ei)), f"trial {i}: expected improvement"
        # step lengths: relative, with the absolute parity bar as the floor (a converged solve ends
        # with steps of 1e-10 and below, made of rounding noise of the gradient)
        close = lambda u, v: abs(math.sqrt(u) - math.sqrt(v)) <= 1e-10 + 1e-9 * math.sqrt(v)
        assert close(st["norm2_cauchy"], t["norm2_cauchy"])
        if st["kind"] != 0:
            assert close(st["norm2_gn"], t["norm2_gn"])
    return worst


@pytest.mark.parametrize("kind", ["dense", "sparse", "products_packed_upper", "products_unpacked"])
def test_oracle_sample_trace_rederived_with_real_lapack(kind):
    """config #1 (the reference's bundled problem), all four modes of check.sh: every trial of the
    oracle's solve against LAPACK dpptrf/dpptrs (dpotrf/dpotrs for unpacked products, as
    dogleg.c:801,889) on the same x, J"""
    P, p0, prm, cookie, cb, k = _sample_setup(kind)
    r, p, tr = oa.oracle_solve(k, p0, 6, 100, 600 if kind == "sparse" else 0, cb, cookie, prm)
    assert r >= 0 and tr.ntrials == 8

    def eval_at(pv):
        x = np.zeros(100)
        J = np.zeros((100, 6))
        P.sample_cb_dense(dptr(np.ascontiguousarray(pv)), dptr(x), dptr(J), None)
        return x, J
    w = _rederive_trace(tr, eval_at, dense_packed=(kind != "products_unpacked"))
    print(f"sample {kind}: {tr.ntrials} trials, worst |step - LAPACK re-derivation| = {w:.2e}")


def test_oracle_dense_trace_rederived_with_real_lapack():
    dp = oa.DenseProblem(M=300, N=24, seed=9, eps=0.4, p0_spread=0.8)
    prm = oa.default_params()
    prm.max_iterations = 12
    prm.trustregion0 = 0.5
    r, p, tr = oa.oracle_solve("dense", dp.p0(), dp.N, dp.M, 0, dp.cb, dp.cookie, prm)
    assert {t["step_type"] for t in tr.trials()} >= {1, 2} or tr.ntrials > 3
    w = _rederive_trace(tr, dp.eval)
    print(f"dense 300x24: {tr.ntrials} trials, worst = {w:.2e}")


@pytest.mark.parametrize("args,prmset", [
    (dict(Nc=4, Np=20, Nobs=60, seed=2, eps=0.4, p0_spread=0.8), dict(max_iterations=15, trustregion0=1.0)),
    (dict(Nc=12, Np=120, Nobs=720, seed=4, eps=0.4, p0_spread=0.6), dict(max_iterations=10, trustregion0=3.0)),
    (dict(Nc=49, Np=900, Nobs=10000, seed=3), dict(max_iterations=6)),
], ids=["ba-tiny", "ba-small", "ba-medium"])
def test_oracle_sparse_trace_rederived_with_superlu(args, prmset):
    """the oracle's sparse Cholesky (the CHOLMOD stand-in) against SuperLU + refinement, trial by trial"""
    prob = oa.BAProblem(args.pop("Nc"), args.pop("Np"), args.pop("Nobs"), **args)
    prm = oa.default_params()
    for k, v in prmset.items():
        setattr(prm, k, v)
    r, p, tr = oa.oracle_solve("sparse", prob.p0(), prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    assert r >= 0
    Jp, Ji = prob.pattern()
    w = _rederive_trace(tr, prob.eval, sparse_pattern=(Jp, Ji, prob.M, prob.N))
    print(f"BA {prob.N} vars: {tr.ntrials} trials, kinds {sorted({t['step_type'] for t in tr.trials()})}, worst = {w:.2e}")


def test_oracle_matches_the_independent_config3_fixture():
    """BASELINE.json config #3 at full size: the oracle's sparse step against the committed
    SuperLU fixture (tests/golden/splu_config3_step.json, make_independent_goldens.py)"""
    O = oa.oracle()
    g = json.load(open(os.path.join(GOLD, "splu_config3_step.json")))
    a = g["problem"]
    prob = oa.BAProblem(a["Nc"], a["Np"], a["Nobs"], seed=a["seed"])
    assert (prob.N, prob.M, prob.nnz) == (g["N"], g["M"], g["nnz"])
    N, M = prob.N, prob.M
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    F = O.orc_sparse_analyze(N, M, iptr(Jp), iptr(Ji))
    work = np.zeros(5 * N)
    o8 = np.zeros(8)
    assert O.orc_step_sparse(F, N, M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(x), dptr(p), 0.0, dptr(work), dptr(o8)) == 0
    O.orc_sparse_free(F)
    gn_ref = np.array([float.fromhex(v) for v in g["gn_hex"]])
    step_ref = np.array([float.fromhex(v) for v in g["step_hex"]])
    dgn = np.linalg.norm(work[2*N:3*N] - gn_ref)
    dst = np.linalg.norm(work[3*N:4*N] - step_ref)
    print(f"config #3 oracle vs SuperLU fixture: |gn diff| = {dgn:.2e}, |step diff| = {dst:.2e}")
    assert dgn <= 1e-10 and dst <= 1e-10
    assert abs(o8[0] - float.fromhex(g["norm2_x"])) <= 1e-12 * o8[0]
    assert abs(o8[1] - float.fromhex(g["norm2_cauchy"])) <= 1e-11 * o8[1]
    assert abs(o8[2] - float.fromhex(g["norm2_gn"])) <= 1e-11 * o8[2]
    assert abs(o8[3] - float.fromhex(g["k"])) <= 1e-10
    assert abs(o8[5] - float.fromhex(g["expected_improvement"])) <= 1e-10 * abs(o8[5])


def test_oracle_dense_solve_matches_the_independent_config2_fixture():
    """BASELINE.json config #2 at full size (dense 50 000 x 2 000): the oracle's dpptrf / dpptrs restatement on BLAS's J'J
    and its Jt*x loop against the committed LAPACK fixture (tests/golden/lapack_config2_step.json) -- the oracle's own
    rank-1 assembly of J'J at this size is 1e11 scalar multiply-adds, a minute: it is pinned on the smaller fixtures"""
    O = oa.oracle()
    g = json.load(open(os.path.join(GOLD, "lapack_config2_step.json")))
    a = g["problem"]
    dp = oa.DenseProblem(M=a["M"], N=a["N"], seed=a["seed"])
    M, N = dp.M, dp.N
    p = dp.p0()
    x, J = dp.eval(p)
    gvec = np.zeros(N)
    O.orc_dense_Jt_x(dptr(gvec), dptr(J), dptr(x), M, N)
    A = J.T @ J
    ap = np.ascontiguousarray(A[np.triu_indices(N)])           # row-major packed upper (dogleg.c:214-220)
    assert O.orc_dpptrf_L(N, dptr(ap)) == 0
    sol = gvec.copy()
    O.orc_dpptrs_L(N, dptr(ap), dptr(sol))
    gn_ref = np.array([float.fromhex(v) for v in g["gn_hex"]])
    d = np.linalg.norm(-sol - gn_ref)
    print(f"config #2 oracle (dpptrf/dpptrs restatement) vs LAPACK fixture: |gn diff| = {d:.2e}")
    assert d <= 1e-10
    assert abs(O.orc_norm2(dptr(x), M) - float.fromhex(g["norm2_x"])) <= 1e-12 * float.fromhex(g["norm2_x"])


def test_oracle_matches_the_independent_config4_fixture():
    """BASELINE.json config #4 at full size (1M x 150k, 15M non-zeros): the oracle's sparse step against the
    committed SuperLU fixture (tests/golden/splu_config4_step.json: every 16th entry of the vectors, their
    norms and sums; make_independent_goldens.py, no oracle and no product in the loop).  ~20 s of CPU."""
    O = oa.oracle()
    g = json.load(open(os.path.join(GOLD, "splu_config4_step.json")))
    a = g["problem"]
    prob = oa.BAProblem(a["Nc"], a["Np"], a["Nobs"], seed=a["seed"])
    assert (prob.N, prob.M, prob.nnz) == (g["N"], g["M"], g["nnz"])
    N, M = prob.N, prob.M
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    F = O.orc_sparse_analyze(N, M, iptr(Jp), iptr(Ji))
    work = np.zeros(5 * N)
    o8 = np.zeros(8)
    assert O.orc_step_sparse(F, N, M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(x), dptr(p), 0.0, dptr(work), dptr(o8)) == 0
    O.orc_sparse_free(F)
    st = g["stride"]
    gn, step = work[2*N:3*N], work[3*N:4*N]
    gn_ref = np.array([float.fromhex(v) for v in g["gn_hex"]])
    step_ref = np.array([float.fromhex(v) for v in g["step_hex"]])
    dgn = np.linalg.norm(gn[::st] - gn_ref)
    dst = np.linalg.norm(step[::st] - step_ref)
    print(f"config #4 oracle vs SuperLU fixture (every {st}th entry): |gn diff| = {dgn:.2e}, |step diff| = {dst:.2e}")
    assert dgn <= 1e-10 and dst <= 1e-10
    # the entries the fixture does not list are pinned through the norms and the sums of the whole vectors
    assert abs(float(gn @ gn) - float.fromhex(g["norm2_gn"])) <= 1e-11 * float(gn @ gn)
    assert abs(float(step @ step) - float.fromhex(g["norm2_step"])) <= 1e-11 * float(step @ step)
    assert abs(float(np.sum(gn)) - float.fromhex(g["sum_gn"])) <= 1e-9 * np.sqrt(N) * np.linalg.norm(gn) / np.sqrt(N)
    assert abs(float(np.sum(step)) - float.fromhex(g["sum_step"])) <= 1e-9 * np.linalg.norm(step)
    assert abs(o8[0] - float.fromhex(g["norm2_x"])) <= 1e-12 * o8[0]
    assert abs(o8[1] - float.fromhex(g["norm2_cauchy"])) <= 1e-11 * o8[1]
    assert abs(o8[3] - float.fromhex(g["k"])) <= 1e-10
    assert abs(o8[5] - float.fromhex(g["expected_improvement"])) <= 1e-10 * abs(o8[5])


def test_oracle_matches_the_independent_config5_fixture():
    """BASELINE.json config #5 at FULL size (5M x 500 001, lambda = 1e-10, cond ~ 1e13): one oracle step against the
    committed independent fixture (tests/golden/splu_config5_step.json, round 5: block elimination of the points +
    SuperLU on the reduced system + iterative refinement with long-double residuals; every 64th entry) -- the oracle's
    sparse Cholesky pinned at the 1e-10 bar on the ill-conditioned configuration too (about 40 s, 3 GB)."""
    g = json.load(open(os.path.join(GOLD, "splu_config5_step.json")))
    a = g["problem"]
    prob = oa.BAProblem(a["Nc"], a["Np"], a["Nobs"], seed=a["seed"], scale_decades=a["scale_decades"], n_zero_cols=a["n_zero_cols"])
    N, M = prob.N, prob.M
    assert (N, M, prob.nnz) == (g["N"], g["M"], g["nnz"])
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    O = oa.oracle()
    lam = float.fromhex(g["lambda"])
    F = O.orc_sparse_analyze(N, M, iptr(Jp), iptr(Ji))
    # (the factorisation at lambda = 0 breaks down on the exactly-zero columns, dogleg.c:667: that is why the step is at 1e-10)
    assert O.orc_sparse_factorize(F, iptr(Jp), iptr(Ji), dptr(Jx), 0.0) != N
    work, o8 = np.zeros(5 * N), np.zeros(8)
    assert O.orc_step_sparse(F, N, M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(x), dptr(p), lam, dptr(work), dptr(o8)) == 0
    O.orc_sparse_free(F)
    gn, step = work[2 * N:3 * N], work[3 * N:4 * N]
    st = g["stride"]
    unhex = lambda lst: np.array([float.fromhex(v) for v in lst])
    dgn = np.linalg.norm(gn[::st] - unhex(g["gn_hex"]))
    dst = np.linalg.norm(step[::st] - unhex(g["step_hex"]))
    print(f"config #5 oracle vs the independent fixture (every {st}th entry): |gn diff| = {dgn:.2e}, |step diff| = {dst:.2e}")
    assert dgn <= 1e-10 and dst <= 1e-10
    assert abs(float(gn @ gn) - float.fromhex(g["norm2_gn"])) <= 1e-11 * float(gn @ gn)
    assert abs(float(step @ step) - float.fromhex(g["norm2_step"])) <= 1e-11 * float(step @ step)
    assert abs(o8[0] - float.fromhex(g["norm2_x"])) <= 1e-12 * o8[0]
    assert abs(o8[1] - float.fromhex(g["norm2_cauchy"])) <= 1e-11 * o8[1]
    assert abs(o8[3] - float.fromhex(g["k"])) <= 1e-10
    assert abs(o8[5] - float.fromhex(g["expected_improvement"])) <= 1e-10 * abs(o8[5])
