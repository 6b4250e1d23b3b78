"""returnContext / dogleg_computeJtJfactorization / dogleg_freeContext as a C user sees them
(reference dogleg.h:269-276, 304-310, 324-328): tests/c/context_harness.c is compiled with gcc
against include/dogleg.h, linked to libdogleg_amd.so, run, and what it read through the context
is checked against numpy: the operating point mirrors, and -- dense -- ctx->factorization_dense
holding the packed factor 'as returned by dpptrf(\'L\', ...)' (dogleg.h:192-194)."""
import os
import subprocess
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "context_harness")
    cmd = ["gcc", "-O1", "-std=gnu11", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "context_harness.c"), "-o", exe,
           "-L", os.path.join(ROOT, "libdogleg_amd"), "-ldogleg_amd",
           "-L", os.path.join(ROOT, "problems"), "-lproblems", "-lm",
           "-Wl,-rpath," + os.path.join(ROOT, "libdogleg_amd"), "-Wl,-rpath," + os.path.join(ROOT, "problems")]
    subprocess.run(cmd, check=True)
    return exe


def _run(exe, mode):
    r = subprocess.run([exe, mode], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = {}
    for line in r.stdout.splitlines():
        k, *v = line.split()
        out[k] = v
    return out


def _f(vals):
    return np.array([float.fromhex(v) for v in vals])


def test_dense_context_and_packed_factor(gpu, tmp_path):
    o = _run(_build(tmp_path), "dense")
    N, M, _ = map(int, o["dims"])
    assert list(map(int, o["ctx"][:3])) == [0, N, M]                      # DOGLEG_DENSE
    assert o["flags"] == ["1", "1", "1"] and o["freed"] == ["1"]
    p, x, J = _f(o["p"]), _f(o["x"]), _f(o["J"]).reshape(M, N)
    assert np.array_equal(p, _f(o["p_out"]))                               # p is overwritten with beforeStep->p
    assert abs(float.fromhex(o["norm2_x"][0]) - x @ x) <= 1e-12 * (x @ x)
    assert float.fromhex(o["result"][0]) == float.fromhex(o["norm2_x"][0])
    assert np.max(np.abs(_f(o["Jt_x"]) - J.T @ x)) <= 1e-11 * max(1.0, np.max(np.abs(J.T @ x)))
    # ctx->factorization_dense: dpptrf('L') on the reference's row-major packed upper JtJ, i.e. the
    # factor U = L^T stored row by row from the diagonal (dogleg.c:214-220, 782-790)
    assert o["have_factorization"][0] == "1"
    lam = float.fromhex(o["have_factorization"][2])
    Lc = np.linalg.cholesky(J.T @ J + lam * np.eye(N))
    packed = np.concatenate([Lc[i:, i] for i in range(N)])
    got = _f(o["factor_packed"])
    assert got.shape == packed.shape
    assert np.max(np.abs(got - packed)) <= 1e-10 * np.max(np.abs(packed))


def test_sparse_context(gpu, tmp_path):
    o = _run(_build(tmp_path), "sparse")
    N, M, nnz = map(int, o["dims"])
    assert list(map(int, o["ctx"][:3])) == [1, N, M]                      # DOGLEG_SPARSE
    assert o["flags"] == ["1", "1", "1"] and o["freed"] == ["1"] and o["factor_handle"] == ["1"]
    x = _f(o["x"])
    Jp, Ji, Jv = np.array(o["Jt_p"], dtype=int), np.array(o["Jt_i"], dtype=int), _f(o["Jt_x_vals"])
    assert Jp[0] == 0 and Jp[-1] == nnz and len(Ji) == nnz
    g = np.zeros(N)
    for r in range(M):
        g[Ji[Jp[r]:Jp[r+1]]] += Jv[Jp[r]:Jp[r+1]] * x[r]
    assert np.max(np.abs(_f(o["Jt_x"]) - g)) <= 1e-11 * max(1.0, np.max(np.abs(g)))
    assert abs(float.fromhex(o["norm2_x"][0]) - x @ x) <= 1e-12 * (x @ x)
    assert np.array_equal(_f(o["p"]), _f(o["p_out"]))
    assert o["have_factorization"][0] == "1"


@pytest.mark.parametrize("presolve", [True, False])
def test_factor_of_a_returned_context_survives_rejected_trials(gpu, presolve, monkeypatch):
    """A sparse solve that ends after SEVERAL rejected trial points (every trial point is made worse than the
    start, the trust region collapses: dogleg.c:1462-1466) hands back a context whose beforeStep reports
    have_factorization; the factor behind it must be the start point's.  The evaluation of a trial point enqueues
    that point's factorisation in the held factor's place (step_prepare), the driver's retry after the rejection
    is dlg_step from the start point's cached vectors: the held factor has to come back BEFORE the step's tail
    clears the spare panel buffer (ADVICE r3 -- it was zeroed there, and solves with the returned factor gave
    garbage).  Checked against the oracle's factorisation of the start point."""
    import ctypes as C
    from libdogleg_amd import capi
    from libdogleg_amd.ctypes_defs import dptr, iptr
    from tests import oracle_api as oa
    if not presolve:
        monkeypatch.setenv("DOGLEG_AMD_NO_PRESOLVE", "1")
    monkeypatch.setenv("DOGLEG_AMD_NO_BACKEND_CACHE", "1")
    prob = oa.BAProblem(12, 120, 720, seed=4, eps=0.4, p0_spread=0.6)
    p0 = prob.p0()
    inner = capi.CB_SPARSE(prob.cb.value)
    nevals = [0]

    @capi.CB_SPARSE
    def cb(p, x, Jt, cookie):
        inner(p, x, Jt, cookie)
        nevals[0] += 1
        if nevals[0] > 1:                       # every trial point: much worse than the start -> rho < 0, rejected
            xv = np.ctypeslib.as_array(x, shape=(prob.M,))
            xv *= 50.0
    L = capi.lib()
    prm = oa.default_params()
    prm.max_iterations = 10
    prm.trustregion0 = 1e3
    prm.trustregion_threshold = 1e-4
    p = p0.copy()
    ctx = C.c_void_p()
    r = L.dogleg_optimize2(dptr(p), prob.N, prob.M, prob.nnz, C.cast(cb, C.c_void_p), prob.cookie, C.byref(prm),
                           C.byref(ctx))
    assert r >= 0 and ctx.value
    assert nevals[0] >= 4, "the scenario needs at least three rejected trial points"
    assert np.array_equal(p, p0)                # nothing was accepted
    be = L.dogleg_amd_backend(ctx)
    rhs = np.cos(0.3 * np.arange(prob.N))
    got, ok = None, 0
    for slot in (0, 1):
        out = np.zeros(prob.N)
        if L.dlg_solve_with_factor(be, slot, dptr(rhs), dptr(out), 1) == 0:
            got, ok = out, ok + 1
    assert ok == 1, "exactly one point holds the factor"
    x0, J0 = prob.eval(p0)
    Jp, Ji = prob.pattern()
    O = oa.oracle()
    F = O.orc_sparse_analyze(prob.N, prob.M, iptr(Jp), iptr(Ji))
    assert O.orc_sparse_factorize(F, iptr(Jp), iptr(Ji), dptr(J0), 0.0) == prob.N
    ref = np.zeros(prob.N)
    O.orc_sparse_solve(F, dptr(rhs), dptr(ref))
    O.orc_sparse_free(F)
    assert np.all(np.isfinite(got))
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)
    L.dogleg_freeContext(C.byref(ctx))
