"""ctypes mirrors of the C structs in include/dogleg.h and include/dlg_trace.h.

Plumbing shared by the product bindings (capi.py) and by the tests' oracle
bindings.  Layouts follow include/dogleg.h, which follows the reference
(/root/reference/dogleg.h:48-210).
"""
import ctypes as C
import numpy as np

DOGLEG_DEBUG_VNLOG_BIT = 30
DOGLEG_DENSE, DOGLEG_SPARSE, DOGLEG_DENSE_PRODUCTS = 0, 1, 2
STEP_NAMES = {0: "cauchy", 1: "gaussnewton", 2: "interpolated"}


class Parameters2(C.Structure):
    """dogleg_parameters2_t (reference dogleg.h:112-152)."""
    _fields_ = [
        ("max_iterations", C.c_int),
        ("dogleg_debug", C.c_int),  # bit0 debug, bit1 JtJ_packed, bit2 JtJ_upper, bit30 vnlog
        ("trustregion0", C.c_double),
        ("trustregion_decrease_factor", C.c_double),
        ("trustregion_decrease_threshold", C.c_double),
        ("trustregion_increase_factor", C.c_double),
        ("trustregion_increase_threshold", C.c_double),
        ("Jt_x_threshold", C.c_double),
        ("update_threshold", C.c_double),
        ("trustregion_threshold", C.c_double),
    ]

    def _bit(self, b, v=None):
        if v is None:
            return bool((self.dogleg_debug >> b) & 1)
        if v:
            self.dogleg_debug |= (1 << b)
        else:
            self.dogleg_debug &= ~(1 << b)

    debug = property(lambda s: s._bit(0), lambda s, v: s._bit(0, v))
    JtJ_packed = property(lambda s: s._bit(1), lambda s, v: s._bit(1, v))
    JtJ_upper = property(lambda s: s._bit(2), lambda s, v: s._bit(2, v))
    debug_vnlog = property(lambda s: s._bit(DOGLEG_DEBUG_VNLOG_BIT),
                           lambda s, v: s._bit(DOGLEG_DEBUG_VNLOG_BIT, v))


class CholmodSparse(C.Structure):
    """cholmod_sparse as far as a dogleg callback touches it."""
    _fields_ = [
        ("nrow", C.c_size_t), ("ncol", C.c_size_t), ("nzmax", C.c_size_t),
        ("p", C.c_void_p), ("i", C.c_void_p), ("nz", C.c_void_p),
        ("x", C.c_void_p), ("z", C.c_void_p),
        ("stype", C.c_int), ("itype", C.c_int), ("xtype", C.c_int),
        ("dtype", C.c_int), ("sorted", C.c_int), ("packed", C.c_int),
    ]


class Trial(C.Structure):
    """dlg_trial_t (include/dlg_trace.h)."""
    _fields_ = [
        ("iteration", C.c_int), ("accepted", C.c_int),
        ("step_type", C.c_int), ("did_step_to_edge", C.c_int),
        ("norm2x_before", C.c_double), ("norm2x_after", C.c_double),
        ("norm2_cauchy", C.c_double), ("norm2_gn", C.c_double),
        ("k_cauchy_to_gn", C.c_double), ("norm2_step", C.c_double),
        ("expected_improvement", C.c_double), ("observed_improvement", C.c_double),
        ("rho", C.c_double),
        ("trustregion_before", C.c_double), ("trustregion_after", C.c_double),
        ("lambda_", C.c_double),
    ]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Trace(C.Structure):
    """dlg_trace_t (include/dlg_trace.h)."""
    _fields_ = [
        ("capacity", C.c_int), ("ntrials", C.c_int),
        ("ncallbacks", C.c_int), ("nstate", C.c_int),
        ("trials", C.POINTER(Trial)),
        ("p_trial", C.POINTER(C.c_double)),
        ("step", C.POINTER(C.c_double)),
    ]


class TraceBuffer:
    """Owns the arrays a dlg_trace_t points at."""

    def __init__(self, nstate, capacity=256):
        self.capacity = capacity
        self.nstate = nstate
        self._trials = (Trial * capacity)()
        self.p_trial = np.zeros((capacity, nstate), dtype=np.float64)
        self.step = np.zeros((capacity, nstate), dtype=np.float64)
        self.c = Trace(capacity, 0, 0, nstate,
                       C.cast(self._trials, C.POINTER(Trial)),
                       self.p_trial.ctypes.data_as(C.POINTER(C.c_double)),
                       self.step.ctypes.data_as(C.POINTER(C.c_double)))

    def byref(self):
        return C.byref(self.c)

    @property
    def ntrials(self):
        return min(self.c.ntrials, self.capacity)

    @property
    def ncallbacks(self):
        return self.c.ncallbacks

    def trials(self):
        return [self._trials[i].asdict() for i in range(self.ntrials)]


CB_SPARSE = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double),
                        C.POINTER(CholmodSparse), C.c_void_p)
CB_DENSE = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double),
                       C.POINTER(C.c_double), C.c_void_p)
CB_PRODUCTS = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double),
                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def dptr(a):
    """double* of a C-contiguous float64 numpy array."""
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int))
