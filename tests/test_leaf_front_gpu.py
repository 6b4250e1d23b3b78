"""Leaf fronts (libdogleg_amd/csrc/sparse_leaf.hip, opt-in with DOGLEG_AMD_LEAF_FRONT=1): JtJ assembly, Jt*x and the
leaf level of the factorisation in ONE kernel.  Same checks as for the separate kernels: every op against the CPU
oracle, whole solves trial by trial, the lambda path, and the one-pass evaluation (dlg_backend_set_speculation)
against the separate calls.  All calls go through the C-ABI.  (DOGLEG_AMD_SYRK_MIN=1: the test problems have fewer
than the 100 leaves from which the two-phase leaf level -- and with it the leaf fronts -- is used by default.)"""
import numpy as np
import pytest

from libdogleg_amd import capi
from tests import oracle_api as oa
from tests.parity import compare_traces
from tests.test_sparse_gpu import _ops_parity

pytestmark = pytest.mark.gpu

VARIANTS = {
    "default": {},
    "row-lists": {"DOGLEG_AMD_LF_LISTS": "1"},          # every strip reads its rows through a list (no arithmetic patterns)
    "no-rider": {"DOGLEG_AMD_LF_NO_RIDER": "1"},        # the global block keeps (split) strips of its own
    "no-stride": {"DOGLEG_AMD_LF_NO_STRIDE": "1"},      # schedules packed back to back (found through the leaf's record)
    "persistent": {"DOGLEG_AMD_LF_WGS": "7"},           # 7 workgroups take the 50 leaves in turn, the next leaf's rows copied into LDS ahead
    "persistent-1": {"DOGLEG_AMD_LF_WGS": "1"},         # one workgroup for all of them
    "per-leaf": {"DOGLEG_AMD_LF_NO_PF": "1"},           # a workgroup per leaf
}


def _on(monkeypatch, extra=None):
    monkeypatch.setenv("DOGLEG_AMD_LEAF_FRONT", "1")
    monkeypatch.setenv("DOGLEG_AMD_SYRK_MIN", "1")
    for k, v in (extra or {}).items():
        monkeypatch.setenv(k, v)


def _is_on(prob):
    Jp, Ji = prob.pattern()
    return capi.symbolic_probe(prob.N, prob.M, Jp, Ji)["leaf_fronts"] == 1


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_ops_match_the_oracle(gpu, variant, monkeypatch):
    _on(monkeypatch, VARIANTS[variant])
    prob = oa.BAProblem(49, 900, 10000, seed=3)
    assert _is_on(prob)
    st, err = _ops_parity(prob)
    print(f"leaf fronts ({variant}):", st, f"|gn diff| = {err:.2e}")


def test_ragged_leaves_and_other_block_widths(gpu, monkeypatch):
    """points with different numbers of observations (the last run of observations is cut), four global variables"""
    _on(monkeypatch)
    prob = oa.BAProblem(40, 800, 8777, g=4, seed=9)
    assert _is_on(prob)
    _ops_parity(prob)


def test_solve_matches_the_oracle_trial_by_trial(gpu, monkeypatch):
    _on(monkeypatch)
    prob = oa.BAProblem(49, 900, 10000, seed=5, eps=0.4, p0_spread=0.6)
    assert _is_on(prob)
    prm = oa.default_params()
    prm.max_iterations = 10
    prm.trustregion0 = 5.0
    p0 = prob.p0()
    ro, po, tro = oa.oracle_solve("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    rg, pg, trg = capi.optimize("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    assert rg >= 0 and ro >= 0
    worst = compare_traces(trg, tro)
    assert np.max(np.abs(pg - po)) <= 1e-10
    print(f"leaf fronts, solve: trials={trg.ntrials} max |step diff|={worst:.3e}")


def test_lambda_path(gpu, monkeypatch):
    """numerically-zero columns in the leaves: the pivot flag of the leaf fronts (a word behind the panels, set at
    assembly time) must reach the factorisation's flag, lambda becomes 1e-10 and the leaves are formed again with it"""
    _on(monkeypatch)
    prob = oa.BAProblem(49, 900, 10000, seed=7, n_zero_cols=3)
    assert _is_on(prob)
    prm = oa.default_params()
    prm.max_iterations = 6
    prm.trustregion0 = 100.0
    p0 = prob.p0()
    ro, po, tro = oa.oracle_solve("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    rg, pg, trg = capi.optimize("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    lam = [t["lambda_"] for t in trg.trials()]
    assert 1e-10 in lam, lam
    compare_traces(trg, tro, step_tol=1e-7)     # lambda=1e-10 systems are ill-conditioned by design


def test_one_pass_evaluation_agrees_with_the_separate_calls(gpu, monkeypatch):
    """dlg_backend_set_speculation: the leaf fronts are formed (and the leaves factored) at evaluation time into the
    second panel buffer; dlg_take_step adopts them.  Against evaluation, factorisation and solve as separate calls
    (where the fronts are formed inside dlg_factorize): same kernel, same sums."""
    _on(monkeypatch)
    prob = oa.BAProblem(49, 900, 10000, seed=4)
    assert _is_on(prob)
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    def run(spec, lam):
        be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
        be.set_pattern(Jp, Ji)
        be.set_speculation(spec)
        be.set_p(0, p)
        be.upload(0, x, Jx)
        n2x, amax = be.eval(0)
        g = be.download(0, capi.VEC_JTX)
        # (lam != 0 with the one-pass evaluation: the fronts were formed at lambda = 0 -- they must NOT be adopted)
        assert be.factorize(0, lam)
        be.solve_gn(0)
        gn = be.download(0, capi.VEC_GN)
        be.close()
        return n2x, g, gn
    n2x0, g0, gn0 = run(False, 0.0)
    n2x1, g1, gn1 = run(True, 0.0)
    assert n2x0 == n2x1
    assert np.max(np.abs(g0 - g1)) <= 1e-12*np.max(np.abs(g0))
    assert np.linalg.norm(gn0 - gn1) <= 1e-12*np.linalg.norm(gn0)
    _, _, gb0 = run(False, 1e-3)
    _, _, gb1 = run(True, 1e-3)
    assert np.linalg.norm(gb0 - gb1) <= 1e-12*np.linalg.norm(gb0)
    assert np.linalg.norm(gb0 - gn0) > 1e-6*np.linalg.norm(gn0)        # (lambda did something)


def test_run_to_run_bitwise_reproducible(gpu, monkeypatch):
    _on(monkeypatch)
    prob = oa.BAProblem(49, 900, 10000, seed=6)
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    res = []
    for _ in range(3):
        be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
        be.set_pattern(Jp, Ji)
        be.set_speculation(True)
        be.set_p(0, p)
        be.upload(0, x, Jx)
        be.eval(0)
        assert be.factorize(0, 0.0)
        be.solve_gn(0)
        res.append((be.download(0, capi.VEC_JTX).copy(), be.download(0, capi.VEC_GN).copy()))
        be.close()
    for g, gn in res[1:]:
        assert np.array_equal(g, res[0][0]) and np.array_equal(gn, res[0][1])
