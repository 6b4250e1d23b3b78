"""ctypes bindings to the CPU oracle (oracle/liboracle.so) and to the problem
library (problems/libproblems.so).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from libdogleg_amd.ctypes_defs import (Parameters2, CholmodSparse, Trace, TraceBuffer,
                                       CB_SPARSE, CB_DENSE, CB_PRODUCTS, dptr, iptr)

from problems import (problems, device_problems, DeviceTwin, fn_addr, BAProblem, DenseProblem)   # noqa: F401  (re-exported: the tests' one-stop import)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = os.path.join(ROOT, "oracle", "liboracle.so")


def build():
    """gcc-build the oracle and the problem library (seconds)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def _stale(lib, *srcs):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


_libs = {}


def oracle():
    if "o" not in _libs:
        if _stale(_ORACLE, os.path.join(ROOT, "oracle", "dogleg_oracle.c"),
                  os.path.join(ROOT, "oracle", "dogleg_oracle.h")):
            build()
        L = C.CDLL(_ORACLE)
        D, I, V = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.orc_optimize_sparse.restype = C.c_double
        L.orc_optimize_sparse.argtypes = [D, C.c_uint, C.c_uint, C.c_uint, V, V,
                                          C.POINTER(Parameters2), C.POINTER(Trace)]
        L.orc_optimize_dense.restype = C.c_double
        L.orc_optimize_dense.argtypes = [D, C.c_uint, C.c_uint, V, V,
                                         C.POINTER(Parameters2), C.POINTER(Trace)]
        L.orc_optimize_dense_products.restype = C.c_double
        L.orc_optimize_dense_products.argtypes = [D, C.c_uint, V, V,
                                                  C.POINTER(Parameters2), C.POINTER(Trace)]
        L.orc_default_parameters.argtypes = [C.POINTER(Parameters2)]
        L.orc_norm2.restype = C.c_double
        L.orc_norm2.argtypes = [D, C.c_uint]
        L.orc_inner.restype = C.c_double
        L.orc_inner.argtypes = [D, D, C.c_uint]
        L.orc_spmv_Jt_x.argtypes = [D, C.c_int, C.c_int, I, I, D, D]
        L.orc_norm2_J_v.restype = C.c_double
        L.orc_norm2_J_v.argtypes = [C.c_int, I, I, D, D]
        L.orc_dense_Jt_x.argtypes = [D, D, D, C.c_int, C.c_int]
        L.orc_dense_norm2_J_v.restype = C.c_double
        L.orc_dense_norm2_J_v.argtypes = [D, D, C.c_int, C.c_int]
        L.orc_xt_Apacked_upper_x.restype = C.c_double
        L.orc_xt_Apacked_upper_x.argtypes = [D, D, C.c_int]
        L.orc_xt_A_x.restype = C.c_double
        L.orc_xt_A_x.argtypes = [D, D, C.c_int]
        L.orc_dense_JtJ_packed_upper.argtypes = [D, D, C.c_int, C.c_int]
        L.orc_dpptrf_L.restype = C.c_int
        L.orc_dpptrf_L.argtypes = [C.c_int, D]
        L.orc_dpptrs_L.argtypes = [C.c_int, D, D]
        L.orc_dpotrf_L.restype = C.c_int
        L.orc_dpotrf_L.argtypes = [C.c_int, D, C.c_int]
        L.orc_dpotrs_L.argtypes = [C.c_int, D, C.c_int, D]
        L.orc_sparse_analyze.restype = V
        L.orc_sparse_analyze.argtypes = [C.c_int, C.c_int, I, I]
        L.orc_sparse_factorize.restype = C.c_long
        L.orc_sparse_factorize.argtypes = [V, I, I, D, C.c_double]
        L.orc_sparse_solve.argtypes = [V, D, D]
        L.orc_sparse_nnzL.restype = C.c_long
        L.orc_sparse_nnzL.argtypes = [V]
        L.orc_sparse_flops.restype = C.c_double
        L.orc_sparse_flops.argtypes = [V]
        L.orc_sparse_free.argtypes = [V]
        L.orc_step_sparse.restype = C.c_int
        L.orc_step_sparse.argtypes = [V, C.c_int, C.c_int, I, I, D, D, D, C.c_double, D, D]
        L.orc_step_dense.restype = C.c_int
        L.orc_step_dense.argtypes = [C.c_int, C.c_int, D, D, D, C.c_double, D, D, D]
        _libs["o"] = L
    return _libs["o"]


def default_params():
    p = Parameters2()
    oracle().orc_default_parameters(C.byref(p))
    return p


def oracle_solve(kind, p0, N, M, nnz, cb, cookie, params=None, capacity=256):
    """Run the oracle.  kind in {'sparse','dense','products'}.
    Returns (norm2x, p_final, TraceBuffer)."""
    L = oracle()
    p = np.array(p0, dtype=np.float64, copy=True)
    tr = TraceBuffer(N, capacity)
    prm = C.byref(params) if params is not None else None
    if kind == "sparse":
        r = L.orc_optimize_sparse(dptr(p), N, M, nnz, cb, cookie, prm, tr.byref())
    elif kind == "dense":
        r = L.orc_optimize_dense(dptr(p), N, M, cb, cookie, prm, tr.byref())
    else:
        r = L.orc_optimize_dense_products(dptr(p), N, cb, cookie, prm, tr.byref())
    return r, p, tr
