"""device-side evaluation.  dogleg_optimize_device2 runs a full solve with the model
evaluated ON the GPU (a dogleg_callback_device_t writing x and the Jacobian values straight into
HBM; reference: the host callback of dogleg.c:1016-1022) and must produce the oracle's iterate
sequence for the same model evaluated by the host callback."""
import ctypes as C
import numpy as np
import pytest

from libdogleg_amd import capi
from tests import oracle_api as oa
from tests.parity import compare_traces

pytestmark = pytest.mark.gpu


def test_device_twin_evaluates_the_same_model(gpu):
    """the device callback used below against the host callback the oracle gets: x and J agree to
    rounding (sin / cos of the device maths library vs glibc)"""
    prob = oa.BAProblem(7, 50, 400, seed=3, eps=0.4)
    twin = oa.DeviceTwin(prob)
    p = prob.p0()
    x, Jx = prob.eval(p)
    dp, dx, dJ = capi.DeviceArray(p), capi.DeviceArray(nbytes=8 * prob.M), capi.DeviceArray(nbytes=8 * prob.nnz)
    fn = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(twin.cb.value)
    fn(dp.ptr, dx.ptr, dJ.ptr, None, twin.cookie)
    assert capi.lib().dlg_device_sync() == 0
    assert np.max(np.abs(dx.numpy() - x)) <= 1e-14 * max(1.0, np.max(np.abs(x)))
    assert np.max(np.abs(dJ.numpy() - Jx)) <= 1e-14 * max(1.0, np.max(np.abs(Jx)))


@pytest.mark.parametrize("shape", [(12, 120, 720), (49, 900, 10000)], ids=["tiny", "medium"])
def test_sparse_device_solve_matches_oracle_trace(gpu, shape):
    prob = oa.BAProblem(*shape, seed=4, eps=0.4, p0_spread=0.6)
    twin = oa.DeviceTwin(prob)
    Jp, Ji = prob.pattern()
    prm = oa.default_params()
    prm.max_iterations = 12
    prm.trustregion0 = 3.0
    p0 = prob.p0()
    ro, po, tro = oa.oracle_solve("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    rg, pg, trg = capi.optimize_device(p0, prob.N, prob.M, prob.nnz, Jp, Ji, twin.cb, twin.cookie, prm)
    assert rg >= 0
    worst = compare_traces(trg, tro)
    assert np.max(np.abs(pg - po)) <= 1e-10
    assert abs(rg - ro) <= 1e-10 * max(1.0, abs(ro))
    # every evaluation ran on the device, none through a host callback; a trial step that ends the solve has had its point
    # evaluated for nothing from inside the step (dlg_backend_set_between: the model's kernels go onto the stream in front of
    # the host's wait for the step) -- never counted, never looked at
    assert trg.ncallbacks == tro.ncallbacks and twin.neval() in (tro.ncallbacks, tro.ncallbacks + 1)
    kinds = {t["step_type"] for t in trg.trials()}
    print(f"device-eval sparse {shape}: {trg.ntrials} trials, step kinds {sorted(kinds)}, max |step diff| {worst:.2e}")


def test_dense_device_solve_matches_oracle_trace(gpu):
    dp = oa.DenseProblem(M=600, N=48, seed=2)
    twin = oa.DeviceTwin(dp)
    prm = oa.default_params()
    prm.max_iterations = 8
    p0 = dp.p0()
    ro, po, tro = oa.oracle_solve("dense", p0, dp.N, dp.M, 0, dp.cb, dp.cookie, prm)
    rg, pg, trg = capi.optimize_device(p0, dp.N, dp.M, 0, None, None, twin.cb, twin.cookie, prm)
    assert rg >= 0
    compare_traces(trg, tro)
    assert np.max(np.abs(pg - po)) <= 1e-10
    assert trg.ncallbacks == tro.ncallbacks and twin.neval() in (tro.ncallbacks, tro.ncallbacks + 1)


def test_device_solve_same_iterates_as_host_callback_solve(gpu):
    """the device entry point and dogleg_optimize2 drive the same kernels: with the same model the
    two traces agree far below the parity tolerance"""
    prob = oa.BAProblem(20, 300, 3000, seed=9, eps=0.3)
    twin = oa.DeviceTwin(prob)
    Jp, Ji = prob.pattern()
    prm = oa.default_params()
    prm.max_iterations = 6
    p0 = prob.p0()
    rh, ph, trh = capi.optimize("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    rd, pd, trd = capi.optimize_device(p0, prob.N, prob.M, prob.nnz, Jp, Ji, twin.cb, twin.cookie, prm)
    compare_traces(trd, trh, step_tol=1e-12, what=("device", "host"))
    assert np.max(np.abs(pd - ph)) <= 1e-12


def test_device_solve_argument_checks(gpu):
    prob = oa.BAProblem(5, 20, 60, seed=1)
    twin = oa.DeviceTwin(prob)
    Jp, Ji = prob.pattern()
    L = capi.lib()
    p = prob.p0()
    # no callback / sparse without a pattern / pattern that disagrees with NJnnz -> -1, no crash
    assert L.dogleg_optimize_device2(capi.dptr(p), prob.N, prob.M, prob.nnz, capi.iptr(Jp), capi.iptr(Ji),
                                     None, None, None, None) < 0
    assert L.dogleg_optimize_device2(capi.dptr(p), prob.N, prob.M, prob.nnz, None, None,
                                     twin.cb, twin.cookie, None, None) < 0
    assert L.dogleg_optimize_device2(capi.dptr(p), prob.N, prob.M, prob.nnz - 1, capi.iptr(Jp), capi.iptr(Ji),
                                     twin.cb, twin.cookie, None, None) < 0


def test_returned_context_of_a_device_solve(gpu):
    """returnContext after a device solve: host mirrors of p / x / Jt_x are filled, the Jacobian
    values stay on the device and dogleg_amd_backend() reaches them and the factor"""
    prob = oa.BAProblem(6, 60, 400, seed=5)
    twin = oa.DeviceTwin(prob)
    Jp, Ji = prob.pattern()
    L = capi.lib()
    p = prob.p0()
    prm = oa.default_params()
    prm.max_iterations = 5
    ctx = C.c_void_p()
    r = L.dogleg_optimize_device2(capi.dptr(p), prob.N, prob.M, prob.nnz, capi.iptr(Jp), capi.iptr(Ji),
                                  twin.cb, twin.cookie, C.byref(prm), C.byref(ctx))
    assert r >= 0 and ctx.value
    be = L.dogleg_amd_backend(ctx)
    assert be
    x, Jx = prob.eval(p)
    # the slot of beforeStep is not known here: one of the two holds the final point's J
    got = []
    for slot in (0, 1):
        out = np.zeros(prob.nnz)
        assert L.dlg_point_download(be, slot, capi.VEC_J, capi.dptr(out), prob.nnz) == 0
        got.append(np.max(np.abs(out - Jx)))
    assert min(got) <= 1e-13 * max(1.0, np.max(np.abs(Jx)))
    L.dogleg_freeContext(C.byref(ctx))
    assert not ctx.value


@pytest.mark.parametrize("overlap", [True, False], ids=["comparison beside the evaluation", "comparison first"])
def test_a_backend_taken_over_serves_the_same_and_another_pattern_of_its_shape(gpu, overlap, monkeypatch):
    """Between solves the library keeps one idle backend with its pattern and schedules (driver.hip: park / take over).  The
    next solve of the SAME shape compares its pattern with the backend's -- on a thread of its own, beside the first
    evaluation, which is made with the backend's schedules (round 5; DOGLEG_AMD_NO_PATTERN_OVERLAP: first, as before).
    Three solves in a row, same (N, M, nnz): pattern A, A again (taken over as it is), then B (another pattern: the
    evaluation made with A's schedules is thrown away, B analysed, the evaluation made again) -- every one the oracle's
    trace, every callback count the oracle's (the model is not evaluated a second time for the thrown-away evaluation)."""
    if overlap:
        monkeypatch.delenv("DOGLEG_AMD_NO_PATTERN_OVERLAP", raising=False)
    else:
        monkeypatch.setenv("DOGLEG_AMD_NO_PATTERN_OVERLAP", "1")
    # (one camera less, two points more: the same number of variables, rows and non-zeros, another pattern)
    probs = [oa.BAProblem(23, 400, 3000, seed=21, eps=0.4, p0_spread=0.5), oa.BAProblem(23, 400, 3000, seed=21, eps=0.4, p0_spread=0.5),
             oa.BAProblem(22, 402, 3000, seed=22, eps=0.4, p0_spread=0.5)]
    assert probs[0].nnz == probs[2].nnz and probs[0].M == probs[2].M and probs[0].N == probs[2].N
    assert not np.array_equal(probs[0].pattern()[1], probs[2].pattern()[1])
    prm = oa.default_params()
    prm.max_iterations = 8
    prm.trustregion0 = 3.0
    for k, prob in enumerate(probs):
        twin = oa.DeviceTwin(prob)
        Jp, Ji = prob.pattern()
        p0 = prob.p0()
        ro, po, tro = oa.oracle_solve("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
        rg, pg, trg = capi.optimize_device(p0, prob.N, prob.M, prob.nnz, Jp, Ji, twin.cb, twin.cookie, prm)
        assert rg >= 0, k
        compare_traces(trg, tro)
        assert np.max(np.abs(pg - po)) <= 1e-10, k
        assert twin.neval() in (tro.ncallbacks, tro.ncallbacks + 1), (k, twin.neval(), tro.ncallbacks)


def test_work_between_a_step_and_its_wait_changes_no_iterate(gpu, monkeypatch):
    """dlg_backend_set_between (driver.hip: the model's kernels for the trial point and the backend's first pass over its
    Jacobian go onto the stream from inside the step, in front of the host's wait for the step's scalars) against
    DOGLEG_AMD_NO_BETWEEN=1 (evaluation enqueued when the host is back, as in rounds 1 - 5): the same kernels on the same
    data -- every trial the same bits, the same final p; without it the model is evaluated exactly as often as the
    reference evaluates it, with it at most once more (the point behind a step that ends the solve).  And where the time of
    a solve goes (DOGLEG_AMD_TIMING=1, dogleg_amd_last_solve_timing): the callback's count is the trace's."""
    monkeypatch.setenv("DOGLEG_AMD_NO_BACKEND_CACHE", "1")
    monkeypatch.setenv("DOGLEG_AMD_TIMING", "1")
    prob = oa.BAProblem(49, 900, 10000, seed=4, eps=0.4, p0_spread=0.6)
    Jp, Ji = prob.pattern()
    prm = oa.default_params()
    prm.max_iterations = 12
    prm.trustregion0 = 3.0
    p0 = prob.p0()
    ro, po, tro = oa.oracle_solve("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    out = {}
    for between in (True, False):
        if between:
            monkeypatch.delenv("DOGLEG_AMD_NO_BETWEEN", raising=False)
        else:
            monkeypatch.setenv("DOGLEG_AMD_NO_BETWEEN", "1")
        twin = oa.DeviceTwin(prob)
        r, p, tr = capi.optimize_device(p0, prob.N, prob.M, prob.nnz, Jp, Ji, twin.cb, twin.cookie, prm)
        tm = capi.last_solve_timing()
        assert r >= 0
        compare_traces(tr, tro)
        assert tm["run_optimizer"][0] > 0.0 and tm["point_eval"][1] == tr.ncallbacks == tro.ncallbacks
        out[between] = (r, p, [t["norm2_step"] for t in tr.trials()], [t["expected_improvement"] for t in tr.trials()], twin.neval())
    assert out[True][0] == out[False][0] and np.array_equal(out[True][1], out[False][1])
    assert out[True][2] == out[False][2] and out[True][3] == out[False][3]
    assert out[False][4] == tro.ncallbacks and out[True][4] in (tro.ncallbacks, tro.ncallbacks + 1)
