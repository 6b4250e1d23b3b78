"""GPU parity tests of the dense / dense-products path against the CPU oracle.
All calls go through the C-ABI (libdogleg_amd.so)."""
import ctypes as C
import numpy as np
import pytest

from libdogleg_amd import capi
from libdogleg_amd.ctypes_defs import dptr
from tests import oracle_api as oa
from tests.parity import compare_traces

pytestmark = pytest.mark.gpu


def _sample(kind):
    P = oa.problems()
    p0 = np.zeros(6)
    P.sample_init(dptr(p0))
    prm = oa.default_params()
    prm.max_iterations = 8
    cookie = None
    cbname = {"dense": "sample_cb_dense", "sparse": "sample_cb_sparse"}.get(kind, "sample_cb_products")
    k = kind
    if kind.startswith("products"):
        k = "products"
        if kind == "products_packed_upper":
            prm.JtJ_packed = True
            prm.JtJ_upper = True
        cookie = C.cast(C.pointer(prm), C.c_void_p)
    return P, p0, prm, cookie, oa.fn_addr(P, cbname), k


@pytest.mark.parametrize("kind", ["dense", "products_packed_upper", "products_unpacked"])
def test_sample_problem_matches_oracle(gpu, kind):
    """reference check.sh:12-14 (`sample --check dense|dense-products-*`) + trial parity"""
    P, p0, prm, cookie, cb, k = _sample(kind)
    ro, po, tro = oa.oracle_solve(k, p0, 6, 100, 0, cb, cookie, prm)
    rg, pg, trg = capi.optimize(k, p0, 6, 100, 0, cb, cookie, prm)
    assert rg >= 0
    # the reference's own assertion (sample.c:424-458)
    assert np.all(np.abs(pg - np.arange(1, 7)) < 5e-2)
    assert abs(rg - ro) <= 1e-9 * max(1, ro)
    worst = compare_traces(trg, tro)
    assert np.max(np.abs(pg - po)) <= 1e-10
    print(f"{kind}: max |step diff| = {worst:.3e}")


def test_dense_synthetic_matches_oracle(gpu):
    prob = oa.DenseProblem(M=3000, N=257, seed=3, eps=0.4, noise=0.02, p0_spread=0.7)
    prm = oa.default_params()
    prm.max_iterations = 12
    prm.trustregion0 = 2.0
    p0 = prob.p0()
    ro, po, tro = oa.oracle_solve("dense", p0, prob.N, prob.M, 0, prob.cb, prob.cookie, prm)
    rg, pg, trg = capi.optimize("dense", p0, prob.N, prob.M, 0, prob.cb, prob.cookie, prm)
    assert rg >= 0 and ro >= 0
    worst = compare_traces(trg, tro)
    assert np.max(np.abs(pg - po)) <= 1e-10
    kinds = {t["step_type"] for t in trg.trials()}
    print(f"dense synthetic: trials={trg.ntrials} step kinds={kinds} max |step diff|={worst:.3e}")


def test_dense_ops_match_oracle(gpu):
    """K1/K3/K4+K5/K6/K7/K8 one at a time against the oracle primitives"""
    O = oa.oracle()
    prob = oa.DenseProblem(M=1500, N=200, seed=5)
    p = prob.p0()
    x, J = prob.eval(p)
    M, N = prob.M, prob.N
    be = capi.Backend(capi.DLG_DENSE, N, M)
    be.set_p(0, p)
    be.upload(0, x, J)
    norm2x, absmax = be.eval(0)
    jtx_ref = np.zeros(N)
    O.orc_dense_Jt_x(dptr(jtx_ref), dptr(J), dptr(x), M, N)
    jtx = be.download(0, capi.VEC_JTX)
    assert np.max(np.abs(jtx - jtx_ref)) <= 1e-12 * max(1, np.max(np.abs(jtx_ref)))
    assert abs(norm2x - O.orc_norm2(dptr(x), M)) <= 1e-12 * norm2x
    assert abs(absmax - np.max(np.abs(jtx_ref))) <= 1e-12 * absmax
    # K3
    n2c = be.cauchy(0)
    g2 = O.orc_norm2(dptr(jtx_ref), N)
    Jg2 = O.orc_dense_norm2_J_v(dptr(J), dptr(jtx_ref), M, N)
    k = -g2 / Jg2
    assert abs(n2c - k * k * g2) <= 1e-12 * n2c
    assert np.max(np.abs(be.download(0, capi.VEC_CAUCHY) - k * jtx_ref)) <= 1e-12 * np.max(np.abs(k * jtx_ref))
    # K4+K5: packed factor against the oracle's dpptrf
    assert be.factorize(0, 0.0)
    ap = np.zeros(N * (N + 1) // 2)
    O.orc_dense_JtJ_packed_upper(dptr(ap), dptr(J), M, N)
    assert O.orc_dpptrf_L(N, dptr(ap)) == 0
    fac = be.factor_dense(N * (N + 1) // 2)
    assert np.max(np.abs(fac - ap)) <= 1e-11 * np.max(np.abs(ap))
    # K6
    n2gn = be.solve_gn(0)
    gn_ref = jtx_ref.copy()
    O.orc_dpptrs_L(N, dptr(ap), dptr(gn_ref))
    gn_ref *= -1
    gn = be.download(0, capi.VEC_GN)
    assert np.linalg.norm(gn - gn_ref) <= 1e-10 * max(1.0, np.linalg.norm(gn_ref))
    assert abs(n2gn - gn_ref @ gn_ref) <= 1e-10 * n2gn
    # K7 interpolated + K8
    tr = 0.5 * (np.sqrt(n2c) + np.sqrt(n2gn))
    if n2c < tr * tr < n2gn:
        n2s, kk, amax, pnew = be.make_step(0, 1, capi.KIND_INTERP, tr)
        a, b = k * jtx_ref, gn_ref
        d = a - b
        l2, negc = d @ d, d @ a
        kref = (negc + np.sqrt(max(0.0, negc * negc - l2 * (a @ a - tr * tr)))) / l2
        sref = a + kref * (b - a)
        assert abs(kk - kref) <= 1e-9
        assert np.linalg.norm(be.download(1, capi.VEC_STEP) - sref) <= 1e-10
        assert np.max(np.abs(pnew - (p + sref))) <= 1e-10
        assert abs(np.sqrt(n2s) - tr) <= 1e-9
        ei = be.expected_improvement(0, 1)
        ei_ref = -2 * (jtx_ref @ sref) - O.orc_dense_norm2_J_v(dptr(J), dptr(sref), M, N)
        assert abs(ei - ei_ref) <= 1e-10 * abs(ei_ref)
    be.close()


def test_dense_lambda_path(gpu):
    """an exactly-zero column of J makes JtJ singular: lambda = 1e-10 then sticks
    (reference dogleg.c:806-815)"""
    M, N = 400, 24
    rng = np.random.default_rng(0)
    J0 = rng.standard_normal((M, N))
    J0[:, 7] = 0.0
    xs = rng.standard_normal(M)

    @capi.CB_DENSE
    def cb(p, x, J, cookie):
        pv = np.ctypeslib.as_array(p, shape=(N,))
        np.ctypeslib.as_array(x, shape=(M,))[:] = J0 @ pv - xs
        np.ctypeslib.as_array(J, shape=(M * N,))[:] = J0.ravel()

    prm = oa.default_params()
    prm.max_iterations = 5
    addr = C.cast(cb, C.c_void_p)
    ro, po, tro = oa.oracle_solve("dense", np.zeros(N), N, M, 0, addr, None, prm)
    rg, pg, trg = capi.optimize("dense", np.zeros(N), N, M, 0, addr, None, prm)
    assert any(t["lambda_"] == 1e-10 for t in trg.trials())
    compare_traces(trg, tro, step_tol=1e-9)


def test_syrk_and_potrf_kernels(gpu):
    """the MFMA SYRK and the blocked Cholesky as stand-alone kernels vs numpy"""
    L = gpu
    rng = np.random.default_rng(1)
    for n, K in ((64, 40), (130, 257), (1100, 333)):
        A = rng.standard_normal((K, n))                  # A[i + k*lda] with lda = n
        dA = capi.DeviceArray(A)
        dC = capi.DeviceArray(nbytes=n * n * 8)
        rc = L.dlg_kernel_syrk_lower(None, dC.ptr, n, dA.ptr, n, n, K, 1.0, None, 0)
        assert rc == 0, L.dlg_last_error()
        assert L.dlg_device_sync() == 0
        ref = A.T @ A
        got = dC.numpy((n, n)).T                         # column-major lower == row-major upper
        low = np.tril(np.ones((n, n), dtype=bool))
        assert np.max(np.abs(got[low] - ref[low])) <= 1e-11 * np.max(np.abs(ref)), (n, K)
        # split-K path through a workspace (overwrite semantics)
        ws = capi.DeviceArray(nbytes=64 << 20)
        dC2 = capi.DeviceArray(nbytes=n * n * 8)
        rc = L.dlg_kernel_syrk_lower(None, dC2.ptr, n, dA.ptr, n, n, K, 1.0, ws.ptr, ws.nbytes)
        assert rc == 0, L.dlg_last_error()
        assert L.dlg_device_sync() == 0
        got2 = dC2.numpy((n, n)).T
        assert np.max(np.abs(got2[low] - ref[low])) <= 1e-11 * np.max(np.abs(ref)), (n, K)
        # potrf of ref + n I
        Sm = ref + n * np.eye(n)                         # symmetric: layout-agnostic
        dS = capi.DeviceArray(Sm)
        dinfo = capi.DeviceArray(np.zeros(1, dtype=np.int32))
        rc = L.dlg_kernel_potrf_lower(None, dS.ptr, n, n, dinfo.ptr)
        assert rc == 0, L.dlg_last_error()
        assert L.dlg_device_sync() == 0
        assert int(dinfo.numpy()[0]) == 0
        Lg = np.tril(dS.numpy((n, n)).T)
        Lref = np.linalg.cholesky(Sm)
        assert np.max(np.abs(Lg - Lref)) <= 1e-11 * np.max(np.abs(Lref)), n


@pytest.mark.parametrize("env", [{"DOGLEG_AMD_NO_OVERLAP": "1"}, {"DOGLEG_AMD_POTRF_STEPS": "1"}, {"DOGLEG_AMD_TRSV_STEPS": "1"}],
                         ids=["no-overlap", "potrf-steps", "trsv-steps"])
def test_dense_stream_variants_match_oracle(gpu, env, monkeypatch):
    """the two-stream variant of the dense path (the Cauchy step beside the factorisation), the step forms of potrf / trsv
    and their single-stream forms give the oracle's Gauss-Newton step"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    O = oa.oracle()
    dp = oa.DenseProblem(M=2500, N=521, seed=4)
    p = dp.p0()
    x, J = dp.eval(p)
    be = capi.Backend(capi.DLG_DENSE, dp.N, dp.M)
    be.set_p(0, p)
    be.upload(0, x, J)
    be.eval(0)
    lam, n2c, n2g = be.cauchy_gauss_newton(0, 0.0)
    gn = be.download(0, capi.VEC_GN)
    N, M = dp.N, dp.M
    dfac, work, o8 = np.zeros(N * (N + 1) // 2), np.zeros(5 * N), np.zeros(8)
    assert O.orc_step_dense(N, M, dptr(J), dptr(x), dptr(p), 0.0, dptr(dfac), dptr(work), dptr(o8)) == 0
    assert np.linalg.norm(gn - work[2*N:3*N]) <= 1e-10
    assert abs(n2c - o8[1]) <= 1e-10 * o8[1]
    be.close()


@pytest.mark.parametrize("N", [130, 521, 1000])
def test_one_launch_potrf_matches_the_step_form_and_reproduces_its_bits(gpu, N, monkeypatch):
    """the whole dense factorisation in one launch (a workgroup per 64 x 64 tile, blocks of L handed over
    through flags: k_potrf_tiles) and both triangular solves in another (k_trsv_tiles): the Gauss-Newton
    step agrees with the step-by-step forms to rounding
    and with numpy's Cholesky solve; a hundred repetitions over two inputs reproduce their bits"""
    dp = oa.DenseProblem(M=3*N, N=N, seed=5)
    p = dp.p0()
    evals = [dp.eval(p), dp.eval(p + 0.01)]
    out = {}
    for mode in ("tiles", "steps"):
        monkeypatch.delenv("DOGLEG_AMD_POTRF_STEPS", raising=False)
        monkeypatch.delenv("DOGLEG_AMD_TRSV_STEPS", raising=False)
        if mode == "steps":
            monkeypatch.setenv("DOGLEG_AMD_POTRF_STEPS", "1")
            monkeypatch.setenv("DOGLEG_AMD_TRSV_STEPS", "1")
        be = capi.Backend(capi.DLG_DENSE, dp.N, dp.M)
        be.set_p(0, p)
        res = []
        for rep in range(100 if mode == "tiles" else 1):
            for x, J in evals:
                be.upload(0, x, J)
                be.eval(0)
                lam, n2g = be.gauss_newton(0, 0.0)
                res.append((n2g, be.download(0, capi.VEC_GN)))
        out[mode] = res
        be.close()
    for k, (n2g, gn) in enumerate(out["tiles"]):
        assert n2g == out["tiles"][k & 1][0] and np.array_equal(gn, out["tiles"][k & 1][1])
    for k in range(2):
        x, J = evals[k]
        ref = -np.linalg.solve(J.reshape(dp.M, dp.N).T @ J.reshape(dp.M, dp.N), J.reshape(dp.M, dp.N).T @ x)
        scale = max(1.0, np.max(np.abs(ref)))
        assert np.max(np.abs(out["tiles"][k][1] - out["steps"][k][1])) <= 1e-11*scale
        assert np.max(np.abs(out["tiles"][k][1] - ref)) <= 1e-9*scale


@pytest.mark.parametrize("trf", [0.5, 1e-3, 1e3])
def test_dense_expected_improvement_behind_the_decision_point(gpu, trf):
    """dlg_backend_set_defer_tail on the dense path: dlg_take_step returns behind the step kernel, the pass over J that forms
    |J step|^2 and the copy of p_new follow on the stream, dlg_step_tail hands the value out (dogleg.c:1427 first uses it
    behind the evaluation of the trial point).  Same kernel, same partial sums -- added in index order on the host where the
    in-line form adds them with k_final's tree: the value agrees to rounding (1e-13 relative asserted), p_new and the step
    bit for bit -- interpolated, Cauchy-to-the-edge and Gauss-Newton steps; the tail fetched at once and behind the next
    evaluation."""
    prob = oa.DenseProblem(M=1500, N=200, seed=7)
    p = prob.p0()
    evals = [prob.eval(p + 0.003*k) for k in range(3)]

    def run(defer, late):
        be = capi.Backend(capi.DLG_DENSE, prob.N, prob.M)
        be.set_p(0, p)
        be.upload(0, *evals[0])
        be.eval(0)
        lam, n2c, n2g = be.cauchy_gauss_newton(0, 0.0)
        tr = trf * (np.sqrt(n2c) + np.sqrt(n2g))
        be.set_defer_tail(defer)
        res, pend, pn = [], None, None
        for rep in range(6):
            be.upload(0, *evals[rep % 3])
            n2x, gmax = be.eval(0)
            if pend is not None:
                pend["ei"] = be.step_tail()
                pend = None
            lam, r, pn = be.take_step(0, 1, tr, 0.0, tail=not late)
            if defer and late:
                assert r["ei"] != r["ei"]
                pend = r
            res.append([n2x, gmax, r, None if (defer and late) else pn.copy(), be.download(1, capi.VEC_STEP)])
        if pend is not None:
            pend["ei"] = be.step_tail()
            res[-1][3] = pn.copy()
        be.close()
        return res
    want = run(False, False)
    for late in (False, True):
        got = run(True, late)
        for i, (a, b) in enumerate(zip(got, want)):
            assert a[0] == b[0] and a[1] == b[1]
            for k in a[2]:
                va, vb = a[2][k], b[2][k]
                assert va == vb or (va != va and vb != vb) or (k == "ei" and abs(va - vb) <= 1e-13 * abs(vb)), (late, i, k, va, vb)
            assert np.isfinite(a[2]["ei"]) and np.array_equal(a[4], b[4])
            if a[3] is not None:
                assert np.array_equal(a[3], b[3])
    print("kinds of step seen:", {q[2]["kind"] for q in want})
