"""The synthetic test problems (problems/problems.c, the SURVEY.md 8d generator and the reference's
sample model re-typed; problems/device_problems.hip: their GPU-resident twins) behind ctypes.
Shared by tests/ and bench.py: inputs only -- nothing here computes what is measured or checked.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from libdogleg_amd.ctypes_defs import dptr, iptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PROBLEMS = os.path.join(ROOT, "problems", "libproblems.so")
_PROBLEMS_DEV = os.path.join(ROOT, "problems", "libproblems_dev.so")
_libs = {}


def build():
    """gcc / hipcc build of the problem libraries (and of the oracle beside them: one Makefile)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def _stale(lib, *srcs):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


def problems():
    if "p" not in _libs:
        if _stale(_PROBLEMS, os.path.join(ROOT, "problems", "problems.c")):
            build()
        L = C.CDLL(_PROBLEMS)
        D, I, V = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.sample_init.argtypes = [D]
        L.sample_set_measurements.argtypes = [D]
        L.sample_get_measurements.argtypes = [D]
        L.synth_ba_create.restype = V
        L.synth_ba_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                      C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
        L.synth_dense_create.restype = V
        L.synth_dense_create.argtypes = [C.c_int, C.c_int, C.c_uint64,
                                         C.c_double, C.c_double, C.c_double]
        L.synth_free.argtypes = [V]
        for f in ("synth_nstate", "synth_nmeas", "synth_nnz", "synth_neval"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [V]
        L.synth_pstar.argtypes = [V, D]
        L.synth_p0.argtypes = [V, D]
        L.synth_pattern.argtypes = [V, I, I]
        L.synth_ba_eval.argtypes = [V, D, D, D]
        L.synth_cb_dense.argtypes = [D, D, D, V]
        L.synth_set_products_layout.argtypes = [C.c_int, C.c_int]
        L.synth_coefs.argtypes = [V, D]
        L.synth_model.argtypes = [V, D, C.POINTER(C.c_uint64)]
        _libs["p"] = L
    return _libs["p"]


def device_problems():
    """problems/libproblems_dev.so: the synthetic problems evaluated on the GPU
    (dogleg_callback_device_t).  Needs a HIP device at call time, not at load time."""
    if "d" not in _libs:
        if _stale(_PROBLEMS_DEV, os.path.join(ROOT, "problems", "device_problems.hip")):
            build()
        L = C.CDLL(_PROBLEMS_DEV)
        D, I, V = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.synth_dev_create_ba.restype = V
        L.synth_dev_create_ba.argtypes = [C.c_int, C.c_int, C.c_int, I, I, D, D, C.c_double, C.c_double, C.c_uint64]
        L.synth_dev_create_dense.restype = V
        L.synth_dev_create_dense.argtypes = [C.c_int, C.c_int, D, C.c_double, C.c_double, C.c_uint64]
        L.synth_dev_free.argtypes = [V]
        L.synth_dev_neval.argtypes = [V]
        L.synth_dev_neval.restype = C.c_int
        _libs["d"] = L
    return _libs["d"]


def fn_addr(lib, name):
    """address of an exported function, as a void* usable as a callback arg."""
    return C.cast(getattr(lib, name), C.c_void_p)


class DeviceTwin:
    """The GPU-resident twin of a BAProblem / DenseProblem: same model, evaluated by a
    dogleg_callback_device_t (problems/device_problems.hip)."""

    def __init__(self, prob):
        self.lib = device_problems()
        en = np.zeros(2)
        seed = C.c_uint64()
        prob.lib.synth_model(prob.h, dptr(en), C.byref(seed))
        pstar = np.zeros(prob.N)
        prob.lib.synth_pstar(prob.h, dptr(pstar))
        if isinstance(prob, BAProblem):
            Jp, Ji = prob.pattern()
            a = np.zeros(prob.nnz)
            prob.lib.synth_coefs(prob.h, dptr(a))
            self.h = self.lib.synth_dev_create_ba(prob.N, prob.M, prob.nnz, iptr(Jp), iptr(Ji), dptr(a),
                                                  dptr(pstar), en[0], en[1], seed.value)
        else:
            self.h = self.lib.synth_dev_create_dense(prob.N, prob.M, dptr(pstar), en[0], en[1], seed.value)
        assert self.h, "device problem creation failed"
        self.cb = fn_addr(self.lib, "synth_cb_device")
        self.cookie = C.c_void_p(self.h)

    def neval(self):
        return self.lib.synth_dev_neval(self.h)

    def close(self):
        if self.h:
            self.lib.synth_dev_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BAProblem:
    """Synthetic block-arrowhead problem (problems.c, SURVEY.md 8d generator)."""

    def __init__(self, Nc, Np, Nobs, g=6, seed=1, eps=0.3, noise=0.01, p0_spread=0.5,
                 scale_decades=0.0, n_zero_cols=0):
        self.lib = problems()
        self.h = self.lib.synth_ba_create(Nc, Np, Nobs, g, seed, eps, noise, p0_spread,
                                          scale_decades, n_zero_cols)
        self.N = self.lib.synth_nstate(self.h)
        self.M = self.lib.synth_nmeas(self.h)
        self.nnz = self.lib.synth_nnz(self.h)
        self.cb = fn_addr(self.lib, "synth_cb_sparse")
        self.cookie = C.c_void_p(self.h)

    def p0(self):
        a = np.zeros(self.N)
        self.lib.synth_p0(self.h, dptr(a))
        return a

    def pstar(self):
        a = np.zeros(self.N)
        self.lib.synth_pstar(self.h, dptr(a))
        return a

    def pattern(self):
        Jp = np.zeros(self.M + 1, dtype=np.int32)
        Ji = np.zeros(self.nnz, dtype=np.int32)
        self.lib.synth_pattern(self.h, iptr(Jp), iptr(Ji))
        return Jp, Ji

    def eval(self, p):
        x = np.zeros(self.M)
        Jx = np.zeros(self.nnz)
        self.lib.synth_ba_eval(self.h, dptr(np.ascontiguousarray(p)), dptr(x), dptr(Jx))
        return x, Jx

    def close(self):
        if self.h:
            self.lib.synth_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DenseProblem:
    def __init__(self, M, N, seed=1, eps=0.3, noise=0.01, p0_spread=0.5):
        self.lib = problems()
        self.h = self.lib.synth_dense_create(M, N, seed, eps, noise, p0_spread)
        self.M, self.N = M, N
        self.cb = fn_addr(self.lib, "synth_cb_dense")
        self.cb_products = fn_addr(self.lib, "synth_cb_products")
        self.cookie = C.c_void_p(self.h)

    def p0(self):
        a = np.zeros(self.N)
        self.lib.synth_p0(self.h, dptr(a))
        return a

    def eval(self, p):
        x = np.zeros(self.M)
        J = np.zeros((self.M, self.N))
        self.lib.synth_cb_dense(dptr(np.ascontiguousarray(p)), dptr(x), dptr(J), self.h)
        return x, J

    def close(self):
        if self.h:
            self.lib.synth_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
