"""Edge cases of the drop-in API on the GPU against the oracle: one variable, one measurement,
under-determined systems (singular JtJ -> the lambda schedule), a start that is already converged,
a huge initial trust region (several rejections), max_iterations = 0/1."""
import ctypes as C
import numpy as np
import pytest

from libdogleg_amd import capi
from libdogleg_amd.ctypes_defs import CholmodSparse
from tests import oracle_api as oa
from tests.parity import compare_traces

pytestmark = pytest.mark.gpu


def _dense_cb(J0, xs, M, N, nonlin=0.0):
    @capi.CB_DENSE
    def cb(p, x, J, cookie):
        pv = np.ctypeslib.as_array(p, shape=(N,)).copy()
        r = J0 @ pv - xs
        np.ctypeslib.as_array(x, shape=(M,))[:] = r + nonlin * np.sin(r)
        Jv = J0 * (1.0 + nonlin * np.cos(r))[:, None]
        np.ctypeslib.as_array(J, shape=(M * N,))[:] = Jv.ravel()
    return cb


def _sparse_cb(J0, xs, M, N, nonlin=0.0):
    """every row lists all N variables (values may be zero)"""
    @capi.CB_SPARSE
    def cb(p, x, Jt, cookie):
        pv = np.ctypeslib.as_array(p, shape=(N,)).copy()
        r = J0 @ pv - xs
        np.ctypeslib.as_array(x, shape=(M,))[:] = r + nonlin * np.sin(r)
        A = Jt.contents
        cp = np.ctypeslib.as_array(C.cast(A.p, C.POINTER(C.c_int)), shape=(M + 1,))
        ri = np.ctypeslib.as_array(C.cast(A.i, C.POINTER(C.c_int)), shape=(M * N,))
        vx = np.ctypeslib.as_array(C.cast(A.x, C.POINTER(C.c_double)), shape=(M * N,))
        cp[:] = np.arange(0, (M + 1) * N, N)
        ri[:] = np.tile(np.arange(N), M)
        vx[:] = (J0 * (1.0 + nonlin * np.cos(r))[:, None]).ravel()
    return cb


def _both(kind, cb, p0, N, M, prm, tol=1e-10):
    addr = C.cast(cb, C.c_void_p)
    nnz = M * N if kind == "sparse" else 0
    ro, po, tro = oa.oracle_solve(kind, p0, N, M, nnz, addr, None, prm)
    rg, pg, trg = capi.optimize(kind, p0, N, M, nnz, addr, None, prm)
    assert (rg < 0) == (ro < 0)
    if ro >= 0:
        compare_traces(trg, tro, step_tol=tol)
        assert np.max(np.abs(pg - po)) <= tol
        assert abs(rg - ro) <= 1e-10 * max(1.0, abs(ro))
    return trg, tro


@pytest.mark.parametrize("kind", ["dense", "sparse"])
def test_one_variable_and_one_measurement(gpu, kind):
    rng = np.random.default_rng(1)
    mk = _dense_cb if kind == "dense" else _sparse_cb
    prm = oa.default_params()
    prm.max_iterations = 10
    for (M, N) in ((5, 1), (1, 1), (7, 2)):
        J0 = rng.standard_normal((M, N)) + 0.5
        xs = rng.standard_normal(M)
        _both(kind, mk(J0, xs, M, N, 0.2), np.full(N, 0.3), N, M, prm)


@pytest.mark.parametrize("kind", ["dense", "sparse"])
def test_underdetermined_system_takes_the_lambda_path(gpu, kind):
    """M < N: JtJ is singular, the factorisation fails until lambda makes it positive definite"""
    rng = np.random.default_rng(2)
    M, N = 6, 10
    J0, xs = rng.standard_normal((M, N)), rng.standard_normal(M)
    mk = _dense_cb if kind == "dense" else _sparse_cb
    prm = oa.default_params()
    prm.max_iterations = 6
    # cond(JtJ + 1e-10 I) ~ 1e11: the two Cholesky factorisations agree to ~1e-5 at best
    trg, tro = _both(kind, mk(J0, xs, M, N), np.zeros(N), N, M, prm, tol=1e-3)
    assert [t["lambda_"] for t in trg.trials()] == [t["lambda_"] for t in tro.trials()]
    assert any(t["lambda_"] > 0 for t in trg.trials())


@pytest.mark.parametrize("kind", ["dense", "sparse"])
def test_converged_start_and_iteration_limits(gpu, kind):
    rng = np.random.default_rng(3)
    M, N = 30, 4
    J0 = rng.standard_normal((M, N))
    pstar = rng.standard_normal(N)
    xs = J0 @ pstar
    mk = _dense_cb if kind == "dense" else _sparse_cb
    prm = oa.default_params()
    prm.max_iterations = 10
    trg, tro = _both(kind, mk(J0, xs, M, N), pstar.copy(), N, M, prm)     # Jt_x = 0 at the start
    assert trg.ncallbacks == tro.ncallbacks == 1 and trg.ntrials == 0
    for it in (0, 1):
        prm2 = oa.default_params()
        prm2.max_iterations = it
        _both(kind, mk(J0, xs + 0.1, M, N, 0.3), np.zeros(N), N, M, prm2)


def test_huge_trust_region_is_cut_down_by_rejections(gpu):
    rng = np.random.default_rng(4)
    M, N = 60, 5
    J0, xs = rng.standard_normal((M, N)), 3.0 * rng.standard_normal(M)
    prm = oa.default_params()
    prm.max_iterations = 25
    prm.trustregion0 = 1e6
    trg, tro = _both("dense", _dense_cb(J0, xs, M, N, 2.5), np.full(N, 4.0), N, M, prm, tol=1e-9)
    assert any(not t["accepted"] for t in trg.trials())


def test_legacy_entry_points_use_the_global_parameters(gpu):
    """dogleg_optimize / dogleg_optimize_dense (reference dogleg.h:278-292) read the process-global
    parameters edited by dogleg_setMaxIterations, dogleg_setInitialTrustregion,
    dogleg_setTrustregionUpdateParameters, dogleg_setThresholds, dogleg_setDebug (dogleg.h:216-257,
    dogleg.c:131-181): a solve through them equals dogleg_optimize*2 with the same values in a struct"""
    L = capi.lib()
    D, U, V = C.POINTER(C.c_double), C.c_uint, C.c_void_p
    L.dogleg_optimize.restype = C.c_double
    L.dogleg_optimize.argtypes = [D, U, U, U, V, V, V]
    L.dogleg_optimize_dense.restype = C.c_double
    L.dogleg_optimize_dense.argtypes = [D, U, U, V, V, V]
    L.dogleg_setMaxIterations.argtypes = [C.c_int]
    L.dogleg_setInitialTrustregion.argtypes = [C.c_double]
    L.dogleg_setTrustregionUpdateParameters.argtypes = [C.c_double] * 4
    L.dogleg_setThresholds.argtypes = [C.c_double] * 3
    L.dogleg_setDebug.argtypes = [C.c_int]
    rng = np.random.default_rng(9)
    M, N = 40, 5
    J0, xs = rng.standard_normal((M, N)), 2.0 * rng.standard_normal(M)
    prm = oa.default_params()
    prm.max_iterations = 7
    prm.trustregion0 = 0.37
    prm.trustregion_decrease_factor, prm.trustregion_decrease_threshold = 0.2, 0.3
    prm.trustregion_increase_factor, prm.trustregion_increase_threshold = 3.0, 0.7
    prm.Jt_x_threshold, prm.update_threshold, prm.trustregion_threshold = 1e-7, 1e-7, 1e-7
    L.dogleg_setMaxIterations(7)
    L.dogleg_setInitialTrustregion(0.37)
    L.dogleg_setTrustregionUpdateParameters(0.2, 0.3, 3.0, 0.7)
    L.dogleg_setThresholds(1e-7, 1e-7, 1e-7)
    L.dogleg_setDebug(0)
    try:
        for kind in ("dense", "sparse"):
            cb = (_dense_cb if kind == "dense" else _sparse_cb)(J0, xs, M, N, 1.2)
            addr = C.cast(cb, C.c_void_p)
            p2 = np.full(N, 0.5)
            r2, p2o, tr2 = capi.optimize(kind, p2, N, M, M * N if kind == "sparse" else 0, addr, None, prm)
            p1 = np.full(N, 0.5)
            if kind == "dense":
                r1 = L.dogleg_optimize_dense(p1.ctypes.data_as(D), N, M, addr, None, None)
            else:
                r1 = L.dogleg_optimize(p1.ctypes.data_as(D), N, M, M * N, addr, None, None)
            assert r1 == r2 and np.array_equal(p1, p2o), (kind, r1, r2)
            assert tr2.ntrials >= 3
    finally:                                               # back to the defaults (dogleg.c:117-128)
        d = oa.default_params()
        L.dogleg_setMaxIterations(d.max_iterations)
        L.dogleg_setInitialTrustregion(d.trustregion0)
        L.dogleg_setTrustregionUpdateParameters(d.trustregion_decrease_factor, d.trustregion_decrease_threshold,
                                                d.trustregion_increase_factor, d.trustregion_increase_threshold)
        L.dogleg_setThresholds(d.Jt_x_threshold, d.update_threshold, d.trustregion_threshold)


@pytest.mark.parametrize("kind", ["dense", "sparse"])
def test_cauchy_step_at_a_singular_point_leaves_lambda_alone(gpu, kind):
    """dogleg.c:1192-1211: when the Cauchy step reaches the edge of the trust region the reference
    never factorises, so a JtJ that is singular only at such a point must not raise the (sticky)
    lambda.  The driver's one-round-trip path factorises speculatively once a step has needed the
    Gauss-Newton step; it has to drop that work -- the lambda column of the trace and every later
    Gauss-Newton step must equal the oracle's.

    The callback keys the Jacobian on the evaluation count: evaluation `sing` returns a tiny J
    (=> a huge Cauchy step, far outside the trust region) with an exactly-zero column."""
    rng = np.random.default_rng(11)
    M, N = 40, 6
    J0 = rng.standard_normal((M, N))
    J0[:, 0] *= 30.0                      # ill-scaled: |cauchy| << |gn| => rejected GN, then interpolation
    xs = rng.standard_normal(M) * 5.0
    state = {"n": 0, "sing": -1}

    def model(pv):
        r = J0 @ pv - xs
        x = r + 0.4 * np.sin(r)
        J = J0 * (1.0 + 0.4 * np.cos(r))[:, None]
        if state["n"] == state["sing"]:
            J = J * 1e-4
            J[:, N - 1] = 0.0
        state["n"] += 1
        return x, J

    if kind == "dense":
        @capi.CB_DENSE
        def cb(p, x, J, cookie):
            xv, Jv = model(np.ctypeslib.as_array(p, shape=(N,)).copy())
            np.ctypeslib.as_array(x, shape=(M,))[:] = xv
            np.ctypeslib.as_array(J, shape=(M * N,))[:] = Jv.ravel()
    else:
        @capi.CB_SPARSE
        def cb(p, x, Jt, cookie):
            xv, Jv = model(np.ctypeslib.as_array(p, shape=(N,)).copy())
            np.ctypeslib.as_array(x, shape=(M,))[:] = xv
            A = Jt.contents
            np.ctypeslib.as_array(C.cast(A.p, C.POINTER(C.c_int)), shape=(M + 1,))[:] = np.arange(0, (M + 1) * N, N)
            np.ctypeslib.as_array(C.cast(A.i, C.POINTER(C.c_int)), shape=(M * N,))[:] = np.tile(np.arange(N), M)
            np.ctypeslib.as_array(C.cast(A.x, C.POINTER(C.c_double)), shape=(M * N,))[:] = Jv.ravel()
    addr = C.cast(cb, C.c_void_p)
    nnz = M * N if kind == "sparse" else 0
    prm = oa.default_params()
    prm.max_iterations = 12
    p0 = np.full(N, 2.0)

    # find an evaluation at which the oracle (a) has already taken a step that needed the GN step
    # and (b) takes a Cauchy step from the point evaluated there
    hit = None
    for sing in range(1, 8):
        state.update(n=0, sing=sing)
        ro, po, tro = oa.oracle_solve(kind, p0, N, M, nnz, addr, None, prm)
        tt = tro.trials()
        # trial t is taken from the point of the last accepted evaluation before it; evaluation e+1
        # is made by trial e
        seen_gn = False
        for e, t in enumerate(tt):
            if e + 1 == sing and t["accepted"] == 1 and seen_gn and e + 1 < len(tt) and tt[e + 1]["step_type"] == 0:
                hit = sing
            if t["step_type"] != 0:
                seen_gn = True
        if hit:
            break
    assert hit is not None, "the test problem no longer produces a Cauchy step after a Gauss-Newton one"
    assert all(t["lambda_"] == 0.0 for t in tt[:hit + 1]), "the oracle must not have factorised at the singular point"

    state.update(n=0, sing=hit)
    ro, po, tro = oa.oracle_solve(kind, p0, N, M, nnz, addr, None, prm)
    state.update(n=0, sing=hit)
    rg, pg, trg = capi.optimize(kind, p0, N, M, nnz, addr, None, prm)
    assert [t["lambda_"] for t in trg.trials()] == [t["lambda_"] for t in tro.trials()]
    compare_traces(trg, tro, step_tol=1e-9)          # the tiny-J point makes one huge, clipped Cauchy step
    assert np.max(np.abs(pg - po)) <= 1e-9


def test_a_hand_off_that_times_out_is_an_error_not_a_wrong_answer(gpu, monkeypatch):
    """The one-launch regions (top of the sparse elimination tree both ways, dense potrf / trsv) hand data
    from workgroup to workgroup through flags; a wait that gives up raises a status word that the host turns
    into DLG_ERR_STATE -- dogleg_optimize* then returns -1.0 -- instead of going on with data that was never
    published.  DOGLEG_AMD_DEBUG_HANDOFF_TIMEOUT (read when the backend is created) makes every wait look for
    an epoch that never comes."""
    prob = oa.BAProblem(49, 900, 10000, seed=9)
    p = prob.p0()
    x, J = prob.eval(p)
    Jp, Ji = prob.pattern()
    dp = oa.DenseProblem(M=1500, N=521, seed=4)
    xd, Jd = dp.eval(dp.p0())

    def sparse_gn():
        be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
        try:
            be.set_pattern(Jp, Ji)
            assert be.schedule()["persist_level0"] >= 0, "the pattern must have a one-launch region"
            be.set_p(0, p); be.upload(0, x, J); be.eval(0)
            return be.gauss_newton(0, 0.0)
        finally:
            be.close()

    def dense_gn():
        be = capi.Backend(capi.DLG_DENSE, dp.N, dp.M)
        try:
            be.set_p(0, dp.p0()); be.upload(0, xd, Jd); be.eval(0)
            return be.gauss_newton(0, 0.0)
        finally:
            be.close()
    ref_s, ref_d = sparse_gn(), dense_gn()
    monkeypatch.setenv("DOGLEG_AMD_DEBUG_HANDOFF_TIMEOUT", "1")
    with pytest.raises(capi.DlgError, match="hand-off"):
        sparse_gn()
    with pytest.raises(capi.DlgError, match="hand-off"):
        dense_gn()
    prm = oa.default_params()
    prm.max_iterations = 5
    r, _, _ = capi.optimize("sparse", p, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    assert r == -1.0
    r, _, _ = capi.optimize("dense", dp.p0(), dp.N, dp.M, 0, dp.cb, dp.cookie, prm)
    assert r == -1.0
    monkeypatch.delenv("DOGLEG_AMD_DEBUG_HANDOFF_TIMEOUT")
    assert sparse_gn() == ref_s and dense_gn() == ref_d          # and nothing sticks to the library


def test_a_pattern_that_changes_between_evaluations_is_refused_when_checked(gpu, monkeypatch):
    """the reference assumes a fixed pattern of Jt (dogleg.c:648-649: analysed once); DOGLEG_AMD_CHECK_PATTERN=1 compares
    every evaluation's pattern with the first one's: the same pattern solves as without the check, a callback that moves
    an entry gets -1.0 instead of a factorisation with a stale schedule"""
    monkeypatch.setenv("DOGLEG_AMD_CHECK_PATTERN", "1")
    monkeypatch.setenv("DOGLEG_AMD_NO_BACKEND_CACHE", "1")
    rng = np.random.default_rng(3)
    M, N = 40, 4
    J0, xs = rng.standard_normal((M, N)), rng.standard_normal(M)
    prm = oa.default_params()
    prm.max_iterations = 6
    cb = _sparse_cb(J0, xs, M, N, 0.3)
    trg, tro = _both("sparse", cb, np.full(N, 1.0), N, M, prm)
    assert trg.ntrials >= 2
    calls = [0]

    @capi.CB_SPARSE
    def moving(p, x, Jt, cookie):
        cb(p, x, Jt, cookie)
        calls[0] += 1
        if calls[0] >= 2:
            A = Jt.contents
            ri = np.ctypeslib.as_array(C.cast(A.i, C.POINTER(C.c_int)), shape=(M * N,))
            ri[0], ri[1] = 1, 0                       # the first row lists its variables in another order
    r, p, tr = capi.optimize("sparse", np.full(N, 1.0), N, M, M * N, C.cast(moving, C.c_void_p), None, prm)
    assert r == -1.0 and calls[0] == 2
