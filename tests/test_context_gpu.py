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
