"""GPU tests against the committed golden vectors and at BASELINE.json sizes."""
import json
import os
import numpy as np
import pytest

from libdogleg_amd import capi
from libdogleg_amd.ctypes_defs import dptr, iptr
from tests import oracle_api as oa

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _unhex(lst):
    return np.array([float.fromhex(v) for v in lst])


def test_gpu_matches_committed_golden_ba_trace(gpu):
    g = json.load(open(os.path.join(GOLD, "oracle_ba_tiny.json")))
    prob = oa.BAProblem(4, 20, 60, seed=2, eps=0.4, p0_spread=0.8)
    prm = oa.default_params()
    prm.max_iterations = 15
    prm.trustregion0 = 1.0
    r, p, tr = capi.optimize("sparse", prob.p0(), prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    assert tr.ncallbacks == g["ncallbacks"] and tr.ntrials == len(g["trials"])
    for i, t in enumerate(g["trials"]):
        rec = tr.trials()[i]
        assert rec["step_type"] == t["step_type"] and rec["accepted"] == t["accepted"]
        assert np.linalg.norm(tr.step[i] - _unhex(t["step_hex"])) <= 1e-10
    assert np.max(np.abs(p - _unhex(g["p_final_hex"]))) <= 1e-10
    assert abs(r - float.fromhex(g["norm2x"])) <= 1e-9 * max(1.0, r)


def test_gpu_matches_committed_golden_dense_trace(gpu):
    g = json.load(open(os.path.join(GOLD, "oracle_dense_small.json")))
    dp = oa.DenseProblem(M=300, N=24, seed=9, eps=0.4, p0_spread=0.8)
    prm = oa.default_params()
    prm.max_iterations = 12
    prm.trustregion0 = 0.5
    r, p, tr = capi.optimize("dense", dp.p0(), dp.N, dp.M, 0, dp.cb, dp.cookie, prm)
    assert tr.ncallbacks == g["ncallbacks"] and tr.ntrials == len(g["trials"])
    for i, t in enumerate(g["trials"]):
        rec = tr.trials()[i]
        assert rec["step_type"] == t["step_type"] and rec["accepted"] == t["accepted"]
        assert np.linalg.norm(tr.step[i] - _unhex(t["step_hex"])) <= 1e-10
    assert np.max(np.abs(p - _unhex(g["p_final_hex"]))) <= 1e-10


def test_gpu_sample_against_reference_trace(gpu):
    """the product itself against the reference's known-answer trace"""
    t = json.load(open(os.path.join(GOLD, "sample_trace.json")))
    P = oa.problems()
    p0 = np.zeros(6)
    P.sample_init(dptr(p0))
    prm = oa.default_params()
    prm.max_iterations = 8
    for kind, cb, nnz in (("sparse", "sample_cb_sparse", 600), ("dense", "sample_cb_dense", 0)):
        r, p, tr = capi.optimize(kind, p0, 6, 100, nnz, oa.fn_addr(P, cb), None, prm)
        assert tr.ncallbacks == t["ncallbacks"] and tr.ntrials == len(t["vnlog"])
        ev = np.array(t["eval_points"])
        for i in range(1, len(ev)):
            assert np.max(np.abs(tr.p_trial[i-1] - ev[i])) < 1e-10, (kind, i)
        assert np.max(np.abs(p - np.array(t["p_final"]))) < 1e-10
        names = {0: "cauchy", 1: "gaussnewton", 2: "interpolated"}
        assert [names[x["step_type"]] for x in tr.trials()] == [row[9] for row in t["vnlog"]]


def _step_parity(prob, lam=0.0, tol=1e-10):
    """one full trial step (the unit bench.py times) on the GPU vs the oracle"""
    O = oa.oracle()
    N, M, nnz = prob.N, prob.M, prob.nnz
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    be = capi.Backend(capi.DLG_SPARSE, N, M, nnz)
    be.set_pattern(Jp, Ji)
    be.set_p(0, p)
    be.upload(0, x, Jx)
    n2x, gmax = be.eval(0)
    n2c = be.cauchy(0)
    ok = be.factorize(0, lam)
    while not ok:
        lam = 1e-10 if lam == 0 else lam * 10
        ok = be.factorize(0, lam)
    n2g = be.solve_gn(0)
    tr = 0.5 * (np.sqrt(n2c) + np.sqrt(n2g))
    n2s, k, amax, pnew = be.make_step(0, 1, capi.KIND_INTERP, tr)
    ei = be.expected_improvement(0, 1)
    step = be.download(1, capi.VEC_STEP)
    F = O.orc_sparse_analyze(N, M, iptr(Jp), iptr(Ji))
    work = np.zeros(5 * N)
    o8 = np.zeros(8)
    assert O.orc_step_sparse(F, N, M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(x), dptr(p), lam, dptr(work), dptr(o8)) == 0
    O.orc_sparse_free(F)
    step_ref = work[3*N:4*N]
    d = np.linalg.norm(step - step_ref)
    assert d <= tol, d
    assert abs(n2x - o8[0]) <= 1e-12 * o8[0]
    assert abs(n2c - o8[1]) <= 1e-10 * o8[1]
    assert abs(n2g - o8[2]) <= 1e-9 * o8[2]
    assert abs(k - o8[3]) <= 1e-9
    assert abs(ei - o8[5]) <= 1e-9 * abs(o8[5])
    assert np.max(np.abs(pnew - work[4*N:5*N])) <= tol
    # size-independent property: the interpolated step sits on the trust-region boundary
    assert abs(np.sqrt(n2s) - tr) <= 1e-9 * tr
    be.close()
    return d, lam


def test_config3_sparse_200k_step_parity(gpu):
    """BASELINE.json configs[2]: 200k meas x 30k params, 3M nnz"""
    d, _ = _step_parity(oa.BAProblem(499, 9000, 100000, seed=11))
    print(f"config #3: |step_gpu - step_oracle| = {d:.3e}")


def test_config4_sparse_1m_step_parity(gpu):
    """BASELINE.json configs[3]: 1M meas x 150k params, 15M nnz (full size, 1 GPU)"""
    d, _ = _step_parity(oa.BAProblem(2499, 45000, 500000, seed=11))
    print(f"config #4: |step_gpu - step_oracle| = {d:.3e}")


def test_ill_conditioned_lambda_step_parity(gpu):
    """configs[4] shape, down-scaled: column scales over 4 decades + exactly-zero columns"""
    d, lam = _step_parity(oa.BAProblem(83, 1500, 25000, seed=13, scale_decades=4.0, n_zero_cols=3), tol=1e-6)
    assert lam >= 1e-10
    print(f"ill-conditioned: lambda={lam:g} |step diff|={d:.3e}")


def test_run_to_run_bitwise_reproducible(gpu):
    """atomics-free, fixed-order reductions: two runs give identical bits"""
    prob = oa.BAProblem(49, 900, 10000, seed=3)
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    outs = []
    for _ in range(2):
        be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
        be.set_pattern(Jp, Ji)
        be.set_p(0, p)
        be.upload(0, x, Jx)
        be.eval(0)
        be.cauchy(0)
        assert be.factorize(0, 0.0)
        be.solve_gn(0)
        outs.append((be.download(0, capi.VEC_JTX), be.download(0, capi.VEC_GN)))
        be.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_dense_config2_shape_downscaled(gpu):
    """configs[1] shape at 1/10 rows: dense 5000 x 2000 ops vs the oracle"""
    O = oa.oracle()
    dp = oa.DenseProblem(M=5000, N=2000, seed=2)
    p = dp.p0()
    x, J = dp.eval(p)
    be = capi.Backend(capi.DLG_DENSE, dp.N, dp.M)
    be.set_p(0, p)
    be.upload(0, x, J)
    be.eval(0)
    n2c = be.cauchy(0)
    assert be.factorize(0, 0.0)
    n2g = be.solve_gn(0)
    gn = be.download(0, capi.VEC_GN)
    A = J.T @ J
    g = J.T @ x
    ref = -np.linalg.solve(A, g)
    assert np.linalg.norm(gn - ref) <= 1e-9 * np.linalg.norm(ref)
    # linearity property of the solve: (JtJ) gn == -g
    assert np.linalg.norm(A @ gn + g) <= 1e-9 * np.linalg.norm(g)
    be.close()
