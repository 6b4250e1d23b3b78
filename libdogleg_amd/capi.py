"""ctypes bindings to libdogleg_amd.so: the dogleg.h API (include/dogleg.h) and
the backend C-ABI (include/dlg_backend.h).  There is no CPU fallback: if the
HIP library is missing this module raises at load time, and every compute
entry point fails with DLG_ERR_NODEVICE when no GPU is present.
"""
import ctypes as C
import os
import numpy as np

from .ctypes_defs import (Parameters2, CholmodSparse, Trace, TraceBuffer,
                          CB_SPARSE, CB_DENSE, CB_PRODUCTS, dptr, iptr)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DLG_TEST_LIB") or os.path.join(HERE, "libdogleg_amd.so")      # (DLG_TEST_LIB: tools only -- the test suite against a variant build, tools/variant_lib.sh)

DLG_OK = 0
DLG_DENSE, DLG_SPARSE, DLG_DENSE_PRODUCTS = 0, 1, 2
FLAG_PACKED, FLAG_UPPER = 1, 2
KIND_CAUCHY, KIND_GN, KIND_INTERP = 0, 1, 2
VEC_P, VEC_X, VEC_JTX, VEC_CAUCHY, VEC_GN, VEC_STEP, VEC_J, VEC_X_OWN, VEC_J_OWN = range(9)

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)

# every symbol include/dlg_backend.h and include/dogleg.h declare
BACKEND_SYMBOLS = [
    "dlg_last_error", "dlg_device_count", "dlg_backend_create", "dlg_backend_destroy",
    "dlg_backend_set_stream", "dlg_backend_get_stream", "dlg_backend_set_shard",
    "dlg_sparse_set_pattern", "dlg_sparse_stats", "dlg_sparse_schedule", "dlg_point_set_p", "dlg_point_upload",
    "dlg_point_upload_products", "dlg_point_bind_device", "dlg_point_eval", "dlg_cauchy",
    "dlg_factorize", "dlg_solve_gn", "dlg_gauss_newton", "dlg_cauchy_gauss_newton", "dlg_make_step",
    "dlg_expected_improvement", "dlg_step", "dlg_take_step", "dlg_solve_with_factor",
    "dlg_point_download", "dlg_factor_download_dense", "dlg_point_device_ptr",
    "dlg_kernel_syrk_lower", "dlg_kernel_potrf_lower", "dlg_probe_mfma_f64", "dlg_probe_mfma_f64_clock", "dlg_probe_mfma_f64_waves",
    "dlg_probe_hbm_copy", "dlg_set_trace", "dlg_mem_alloc", "dlg_mem_free", "dlg_host_alloc",
    "dlg_host_free", "dlg_mem_upload",
    "dlg_mem_download", "dlg_mem_zero", "dlg_device_sync", "dlg_sparse_symbolic_probe",
    "dlg_backend_set_profiling", "dlg_backend_get_profile", "dlg_backend_get_profile_early",
    "dlg_backend_set_allreduce", "dlg_backend_set_partition", "dlg_partition_rows", "dlg_partition_stats",
    "dlg_sparse_partition_probe", "dlg_rccl_unique_id", "dlg_backend_init_rccl", "dlg_backend_set_rccl",
    "dlg_backend_comm_size", "dlg_backend_has_rccl", "dlg_backend_set_noop_comm", "dlg_solve_multi", "dlg_pseudoinverse_chunk", "dlg_backend_set_speculation", "dlg_backend_set_defer_tail", "dlg_step_tail",
    "dlg_step_tail_pending", "dlg_backend_ei_source", "dlg_backend_set_between", "dlg_backend_between_redone", "dlg_point_eval_early",
    "dlg_backend_share_rccl", "dlg_point_gather_device", "dlg_backend_reset", "dlg_backend_device",
    "dlg_sparse_pattern_matches", "dlg_sparse_drop_pattern", "dlg_sparse_region_probe", "dlg_run_steps", "dlg_backend_time_allreduce",
]
PROF_NAMES = ["K1_jtx", "K3K8_norm2Jv", "K4_kernel", "K4_total", "K5_factor", "K6_solve", "K7_step", "vec"]
DOGLEG_SYMBOLS = [
    "dogleg_getDefaultParameters", "dogleg_setMaxIterations",
    "dogleg_setTrustregionUpdateParameters", "dogleg_setDebug", "dogleg_setInitialTrustregion",
    "dogleg_setThresholds", "dogleg_optimize", "dogleg_optimize2", "dogleg_optimize_dense",
    "dogleg_optimize_dense2", "dogleg_optimize_dense_products", "dogleg_computeJtJfactorization",
    "dogleg_freeContext", "dogleg_testGradient", "dogleg_testGradient_dense",
    "dogleg_testGradient_dense_products",
    "dogleg_optimize_device2", "dogleg_amd_backend", "dogleg_amd_point_slot",
    "dogleg_amd_set_communicator", "dogleg_amd_set_allreduce", "dogleg_amd_clear_communicator",
    "dogleg_amd_rccl_unique_id", "dogleg_amd_rank", "dogleg_amd_release_cache",
    "dogleg_amd_id_file_publish", "dogleg_amd_id_file_wait", "dogleg_amd_last_solve_timing",
]

_lib = None


class DlgError(RuntimeError):
    pass


def lib():
    """Load libdogleg_amd.so (building it is __graft_entry__.build()'s job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DlgError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    D, I, V = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    L.dlg_last_error.restype = C.c_char_p
    L.dlg_device_count.restype = C.c_int
    L.dlg_backend_create.argtypes = [C.POINTER(V), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.dlg_backend_destroy.argtypes = [V]
    L.dlg_backend_destroy.restype = None
    L.dlg_backend_set_stream.argtypes = [V, V]
    L.dlg_backend_get_stream.argtypes = [V]
    L.dlg_backend_get_stream.restype = V
    L.dlg_backend_set_shard.argtypes = [V, C.c_int, C.c_int, V, V]
    L.dlg_backend_set_allreduce.argtypes = [V, V, V]
    L.dlg_backend_set_speculation.argtypes = [V, C.c_int]
    L.dlg_backend_set_defer_tail.argtypes = [V, C.c_int]
    L.dlg_step_tail.argtypes = [V, C.POINTER(C.c_double)]
    L.dlg_step_tail_pending.argtypes = [V]
    L.dlg_backend_ei_source.argtypes = [V, I, D]
    L.dlg_backend_set_partition.argtypes = [V, C.c_int, C.c_int]
    L.dlg_partition_rows.argtypes = [V, I, C.POINTER(I)]
    L.dlg_partition_stats.argtypes = [V, C.POINTER(C.c_long), C.c_int]
    L.dlg_sparse_partition_probe.argtypes = [C.c_int, C.c_int, I, I, C.c_int, C.c_int, C.POINTER(C.c_long), C.c_int, C.c_char_p]
    L.dlg_rccl_unique_id.argtypes = [V]
    L.dlg_backend_init_rccl.argtypes = [V, C.c_int, C.c_int, V]
    L.dlg_backend_set_rccl.argtypes = [V, V]
    L.dlg_backend_comm_size.argtypes = [V, I]
    L.dlg_backend_has_rccl.argtypes = [V]
    L.dlg_backend_set_noop_comm.argtypes = [V, C.c_int]
    L.dlg_backend_get_profile_early.argtypes = [V, D, C.POINTER(C.c_long), C.c_int]
    L.dlg_sparse_set_pattern.argtypes = [V, I, I]
    L.dlg_sparse_stats.argtypes = [V, C.POINTER(C.c_long), C.POINTER(C.c_long), I, I, D]
    L.dlg_sparse_schedule.argtypes = [V, I, I, I]
    L.dlg_point_set_p.argtypes = [V, C.c_int, D]
    L.dlg_point_upload.argtypes = [V, C.c_int, D, D]
    L.dlg_point_upload_products.argtypes = [V, C.c_int, C.c_double, D, D]
    L.dlg_point_bind_device.argtypes = [V, C.c_int, V, V]
    L.dlg_point_eval.argtypes = [V, C.c_int, D, D]
    L.dlg_cauchy.argtypes = [V, C.c_int, D]
    L.dlg_factorize.argtypes = [V, C.c_int, C.c_double, I]
    L.dlg_solve_gn.argtypes = [V, C.c_int, D]
    L.dlg_gauss_newton.argtypes = [V, C.c_int, D, D]
    L.dlg_cauchy_gauss_newton.argtypes = [V, C.c_int, D, D, D]
    L.dlg_make_step.argtypes = [V, C.c_int, C.c_int, C.c_int, C.c_double, D, D, D, D]
    L.dlg_step.argtypes = [V, C.c_int, C.c_int, C.c_int, C.c_double, D, D, D, D, D]
    L.dlg_take_step.argtypes = [V, C.c_int, C.c_int, C.c_double, D, D, D]
    L.dlg_solve_with_factor.argtypes = [V, C.c_int, D, D, C.c_int]
    L.dlg_solve_multi.argtypes = [V, C.c_int, D, D, C.c_int]
    L.dlg_pseudoinverse_chunk.argtypes = [V, C.c_int, C.c_int, C.c_int, D]
    L.dlg_expected_improvement.argtypes = [V, C.c_int, C.c_int, D]
    L.dlg_point_download.argtypes = [V, C.c_int, C.c_int, D, C.c_size_t]
    L.dlg_factor_download_dense.argtypes = [V, D, C.c_size_t]
    L.dlg_point_device_ptr.argtypes = [V, C.c_int, C.c_int]
    L.dlg_point_device_ptr.restype = V
    L.dlg_kernel_syrk_lower.argtypes = [V, V, C.c_int, V, C.c_int, C.c_int, C.c_int, C.c_double, V, C.c_size_t]
    L.dlg_kernel_potrf_lower.argtypes = [V, V, C.c_int, C.c_int, V]
    L.dlg_probe_mfma_f64.argtypes = [D]
    L.dlg_probe_hbm_copy.argtypes = [D]
    L.dlg_set_trace.argtypes = [V]
    L.dlg_set_trace.restype = None
    L.dlg_mem_alloc.argtypes = [C.c_size_t]
    L.dlg_mem_alloc.restype = V
    L.dlg_mem_free.argtypes = [V]
    L.dlg_mem_free.restype = None
    L.dlg_host_alloc.argtypes = [C.c_size_t]
    L.dlg_host_alloc.restype = V
    L.dlg_host_free.argtypes = [V]
    L.dlg_host_free.restype = None
    L.dlg_mem_upload.argtypes = [V, V, C.c_size_t]
    L.dlg_mem_download.argtypes = [V, V, C.c_size_t]
    L.dlg_mem_zero.argtypes = [V, C.c_size_t]
    L.dlg_backend_set_profiling.argtypes = [V, C.c_int]
    L.dlg_backend_get_profile.argtypes = [V, D, C.POINTER(C.c_long), C.c_int]
    L.dlg_sparse_symbolic_probe.argtypes = [C.c_int, C.c_int, I, I, C.c_int, C.c_int,
                                            C.POINTER(C.c_long), C.c_int, I]
    # dogleg.h
    PP = C.POINTER(Parameters2)
    L.dogleg_getDefaultParameters.argtypes = [PP]
    L.dogleg_getDefaultParameters.restype = None
    L.dogleg_optimize2.restype = C.c_double
    L.dogleg_optimize2.argtypes = [D, C.c_uint, C.c_uint, C.c_uint, V, V, PP, V]
    L.dogleg_optimize.restype = C.c_double
    L.dogleg_optimize.argtypes = [D, C.c_uint, C.c_uint, C.c_uint, V, V, V]
    L.dogleg_optimize_dense2.restype = C.c_double
    L.dogleg_optimize_dense2.argtypes = [D, C.c_uint, C.c_uint, V, V, PP, V]
    L.dogleg_optimize_dense.restype = C.c_double
    L.dogleg_optimize_dense.argtypes = [D, C.c_uint, C.c_uint, V, V, V]
    L.dogleg_optimize_dense_products.restype = C.c_double
    L.dogleg_optimize_dense_products.argtypes = [D, C.c_uint, V, V, PP, V]
    L.dogleg_freeContext.argtypes = [V]
    L.dogleg_freeContext.restype = None
    L.dogleg_optimize_device2.restype = C.c_double
    L.dogleg_optimize_device2.argtypes = [D, C.c_uint, C.c_uint, C.c_uint, I, I, V, V, PP, V]
    L.dogleg_amd_backend.restype = V
    L.dogleg_amd_backend.argtypes = [V]
    L.dogleg_amd_point_slot.restype = C.c_int
    L.dogleg_amd_point_slot.argtypes = [V, V]
    L.dogleg_amd_set_communicator.argtypes = [C.c_int, C.c_int, C.c_int, V]
    L.dogleg_amd_set_allreduce.argtypes = [C.c_int, C.c_int, C.c_int, V, V]
    L.dogleg_amd_clear_communicator.restype = None
    L.dogleg_amd_rccl_unique_id.argtypes = [V]
    L.dogleg_amd_rank.argtypes = [V, I]
    L.dogleg_amd_id_file_publish.argtypes = [C.c_char_p, V, C.c_char_p]
    L.dogleg_amd_id_file_wait.argtypes = [C.c_char_p, V, C.c_char_p, C.c_int]
    L.dlg_backend_share_rccl.argtypes = [V, V]
    L.dlg_sparse_region_probe.argtypes = [C.c_int, C.c_int, I, I, C.c_int, C.POINTER(C.c_long), C.c_int]
    L.dlg_run_steps.argtypes = [V, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(V), C.POINTER(V), C.c_int, C.c_double, C.c_double, D, I]
    L.dlg_backend_reset.argtypes = [V]
    L.dlg_backend_device.argtypes = [V]
    L.dlg_sparse_pattern_matches.argtypes = [V, I, I]
    L.dlg_sparse_drop_pattern.argtypes = [V]
    L.dogleg_amd_release_cache.restype = None
    L.dogleg_amd_last_solve_timing.argtypes = [D, I]
    L.dlg_point_gather_device.argtypes = [V, C.c_int, V, V, I]
    L.dogleg_setMaxIterations.argtypes = [C.c_int]
    L.dogleg_setDebug.argtypes = [C.c_int]
    L.dogleg_setInitialTrustregion.argtypes = [C.c_double]
    L.dogleg_setThresholds.argtypes = [C.c_double, C.c_double, C.c_double]
    L.dogleg_setTrustregionUpdateParameters.argtypes = [C.c_double] * 4
    _lib = L
    return L


def _ck(rc, what=""):
    if rc != DLG_OK:
        raise DlgError(f"{what} failed (code {rc}): {lib().dlg_last_error().decode()}")


def default_parameters():
    p = Parameters2()
    lib().dogleg_getDefaultParameters(C.byref(p))
    return p


def optimize(kind, p0, N, M, nnz, cb, cookie, params=None, capacity=256, trace=True):
    """dogleg_optimize2 / _dense2 / _dense_products with a per-trial trace (trace=False: without, as a user calls it).
    kind in {'sparse','dense','products'}; cb is a function address (c_void_p).
    Returns (norm2x, p_final, TraceBuffer)."""
    L = lib()
    p = np.array(p0, dtype=np.float64, copy=True)
    tr = TraceBuffer(N, capacity) if trace else None
    prm = C.byref(params) if params is not None else None
    L.dlg_set_trace(C.cast(tr.byref(), C.c_void_p) if trace else None)
    try:
        if kind == "sparse":
            r = L.dogleg_optimize2(dptr(p), N, M, nnz, cb, cookie, prm, None)
        elif kind == "dense":
            r = L.dogleg_optimize_dense2(dptr(p), N, M, cb, cookie, prm, None)
        else:
            r = L.dogleg_optimize_dense_products(dptr(p), N, cb, cookie, prm, None)
    finally:
        L.dlg_set_trace(None)
    return r, p, tr


def optimize_device(p0, N, M, nnz, Jp, Ji, cb, cookie, params=None, capacity=256, trace=True):
    """dogleg_optimize_device2 (device-side evaluation) with a per-trial trace.  nnz == 0: dense
    (Jp, Ji ignored).  cb: address of a dogleg_callback_device_t.  Returns (norm2x, p_final, TraceBuffer).
    trace=False: no per-trial record (every record downloads the step vector: a test feature that costs a
    synchronisation a trial) -- what a user's call does; the third value is None then."""
    L = lib()
    p = np.array(p0, dtype=np.float64, copy=True)
    tr = TraceBuffer(N, capacity) if trace else None
    prm = C.byref(params) if params is not None else None
    if nnz > 0:
        Jp = np.ascontiguousarray(Jp, dtype=np.int32)
        Ji = np.ascontiguousarray(Ji, dtype=np.int32)
    L.dlg_set_trace(C.cast(tr.byref(), C.c_void_p) if trace else None)
    try:
        r = L.dogleg_optimize_device2(dptr(p), N, M, nnz, iptr(Jp) if nnz > 0 else None,
                                      iptr(Ji) if nnz > 0 else None, cb, cookie, prm, None)
    finally:
        L.dlg_set_trace(None)
    return r, p, tr


SYM_STAT_NAMES = ["var_blocks", "supernodes", "levels", "nnz_JtJ_lower", "nnz_L", "panel_doubles",
                  "factor_flops", "max_panel", "asm_tasks", "update_items", "relpos", "out_blocks",
                  "contribs", "update_subtasks", "solve_scratch", "jtx_tasks", "asm_mfma_tasks",
                  "asm_kgroups", "asm_shapes"]


def symbolic_probe(N, M, Jp, Ji, row0=0, row1=None, want_perm=False):
    """Host-only symbolic analysis of a Jt pattern: dict of statistics (+ perm)."""
    L = lib()
    Jp = np.ascontiguousarray(Jp, dtype=np.int32)
    Ji = np.ascontiguousarray(Ji, dtype=np.int32)
    st = (C.c_long * len(SYM_STAT_NAMES))()
    perm = np.zeros(N, dtype=np.int32) if want_perm else None
    _ck(L.dlg_sparse_symbolic_probe(N, M, iptr(Jp), iptr(Ji), row0, M if row1 is None else row1,
                                    st, len(SYM_STAT_NAMES), iptr(perm) if want_perm else None),
        "symbolic probe")
    d = {k: st[i] for i, k in enumerate(SYM_STAT_NAMES)}
    return (d, perm) if want_perm else d


PART_STAT_NAMES = ["cut_level", "supernodes_above_cut", "supernodes_mine", "rows_mine", "reduced_doubles",
                   "panel_doubles", "nnz_mine"]


def partition_probe(N, M, Jp, Ji, rank, nranks):
    """Host-only: the subtree partition of a Jt pattern as rank `rank` of `nranks` sees it.
    Returns (dict of statistics, bool array row_is_mine[M])."""
    L = lib()
    Jp = np.ascontiguousarray(Jp, dtype=np.int32)
    Ji = np.ascontiguousarray(Ji, dtype=np.int32)
    st = (C.c_long * len(PART_STAT_NAMES))()
    own = np.zeros(M, dtype=np.uint8)
    _ck(L.dlg_sparse_partition_probe(N, M, iptr(Jp), iptr(Ji), rank, nranks, st, len(PART_STAT_NAMES),
                                     own.ctypes.data_as(C.c_char_p)), "partition probe")
    return {k: st[i] for i, k in enumerate(PART_STAT_NAMES)}, own.astype(bool)


REGION_STAT_NAMES = ["level0", "supernodes", "workgroups", "lds_bytes", "sliced_workgroups", "hbm_update_matrices"]


def region_probe(N, M, Jp, Ji, ncu=256):
    """Host-only: the one-launch region of the factorisation of a pattern on a chip with `ncu` CUs, checked
    (raises DlgError on a violated invariant).  Returns a dict of REGION_STAT_NAMES."""
    L = lib()
    Jp = np.ascontiguousarray(Jp, dtype=np.int32)
    Ji = np.ascontiguousarray(Ji, dtype=np.int32)
    st = (C.c_long * len(REGION_STAT_NAMES))()
    _ck(L.dlg_sparse_region_probe(N, M, iptr(Jp), iptr(Ji), ncu, st, len(REGION_STAT_NAMES)), "region probe")
    return {k: st[i] for i, k in enumerate(REGION_STAT_NAMES)}


def rccl_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 creates it and hands it to the others)"""
    buf = (C.c_char * 128)()
    _ck(lib().dlg_rccl_unique_id(C.cast(buf, C.c_void_p)), "rccl_unique_id")
    return bytes(buf)


class DeviceArray:
    """A hipMalloc'ed buffer filled from / read back into numpy (harness helper)."""

    def __init__(self, arr=None, nbytes=None, dtype=np.float64):
        L = lib()
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            nbytes, dtype = arr.nbytes, arr.dtype
        self.nbytes, self.dtype = nbytes, np.dtype(dtype)
        self.ptr = L.dlg_mem_alloc(nbytes)
        if not self.ptr:
            raise DlgError(L.dlg_last_error().decode())
        if arr is not None:
            _ck(L.dlg_mem_upload(self.ptr, arr.ctypes.data, nbytes), "upload")
        else:
            _ck(L.dlg_mem_zero(self.ptr, nbytes), "memset")

    def numpy(self, shape=None):
        out = np.zeros(self.nbytes // self.dtype.itemsize, dtype=self.dtype)
        _ck(lib().dlg_mem_download(out.ctypes.data, self.ptr, self.nbytes), "download")
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            lib().dlg_mem_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Backend:
    """Thin OO wrapper over the dlg_backend C-ABI (include/dlg_backend.h)."""

    def __init__(self, solve_type, N, M, nnz=0, flags=0, device=-1):
        self.L = lib()
        self.h = C.c_void_p()
        self.N, self.M, self.nnz, self.type = N, M, nnz, solve_type
        _ck(self.L.dlg_backend_create(C.byref(self.h), solve_type, N, M, nnz, flags, device),
            "dlg_backend_create")
        self._keep = []
        self._pnew_ptr = None
        self._pnew = None

    def close(self):
        if self.h:
            self.L.dlg_backend_destroy(self.h)
            self.h = C.c_void_p()
        if self._pnew_ptr:
            self.L.dlg_host_free(self._pnew_ptr)
            self._pnew_ptr, self._pnew = None, None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_profiling(self, on=True, only=None, every=1):
        """only: names of the phases to time (PROF_NAMES), default all; every: time every n-th occurrence only"""
        v = (1 if on else 0) if only is None else sum(2 << PROF_NAMES.index(n) for n in only)
        if v and every > 1:
            v |= min(int(every), 255) << 16
        _ck(self.L.dlg_backend_set_profiling(self.h, v), "set_profiling")

    def profile(self):
        """{phase: (total_ms, launches)} since set_profiling(True)"""
        n = len(PROF_NAMES)
        ms = (C.c_double * n)()
        cnt = (C.c_long * n)()
        _ck(self.L.dlg_backend_get_profile(self.h, ms, cnt, n), "get_profile")
        return {PROF_NAMES[i]: (ms[i], cnt[i]) for i in range(n)}

    def profile_early(self):
        """{phase: (total_ms, launches)} of the launches that returned early behind a failed factorisation"""
        n = len(PROF_NAMES)
        ms = (C.c_double * n)()
        cnt = (C.c_long * n)()
        _ck(self.L.dlg_backend_get_profile_early(self.h, ms, cnt, n), "get_profile_early")
        return {PROF_NAMES[i]: (ms[i], cnt[i]) for i in range(n)}

    def set_stream(self, stream_ptr):
        _ck(self.L.dlg_backend_set_stream(self.h, C.c_void_p(stream_ptr)), "set_stream")

    def set_shard(self, row0, row1, fn=None):
        cb = ALLREDUCE_FN(fn) if fn is not None else None
        self._allreduce_cb = cb                 # must outlive the backend: the C side keeps the pointer
        _ck(self.L.dlg_backend_set_shard(self.h, row0, row1,
                                         C.cast(cb, C.c_void_p) if cb else None, None), "set_shard")

    def set_speculation(self, on=True):
        """assemble JtJ beside Jt*x at every eval (for callers that expect to factorise the point)"""
        _ck(self.L.dlg_backend_set_speculation(self.h, 1 if on else 0), "set_speculation")

    def set_defer_tail(self, on=True):
        """dlg_take_step returns before the expected improvement's pass over J (K8): step_tail() has the value"""
        _ck(self.L.dlg_backend_set_defer_tail(self.h, 1 if on else 0), "set_defer_tail")

    def step_tail(self):
        """the expected improvement of the last step taken with set_defer_tail (and its p_new complete)"""
        v = C.c_double()
        _ck(self.L.dlg_step_tail(self.h, C.byref(v)), "step_tail")
        return v.value

    def step_tail_pending(self):
        return bool(self.L.dlg_step_tail_pending(self.h))

    def ei_source(self):
        """(the last expected improvement came from the solved system -- no pass over J --, pivot ratio of the factor)"""
        f, r = C.c_int(), C.c_double()
        _ck(self.L.dlg_backend_ei_source(self.h, C.byref(f), C.byref(r)), "ei_source")
        return bool(f.value), r.value

    def set_allreduce(self, fn):
        """host-synchronous sum-all-reduce hook (fallback / logical ranks on one device)"""
        cb = ALLREDUCE_FN(fn) if fn is not None else None
        self._allreduce_cb = cb
        _ck(self.L.dlg_backend_set_allreduce(self.h, C.cast(cb, C.c_void_p) if cb else None, None), "set_allreduce")

    def set_partition(self, rank, nranks):
        """sparse: subtree partition over nranks ranks (before set_pattern)"""
        _ck(self.L.dlg_backend_set_partition(self.h, rank, nranks), "set_partition")

    def partition_rows(self):
        """the measurement rows this rank holds (ascending); x / J values are uploaded for these"""
        n = C.c_int()
        rows = C.POINTER(C.c_int)()
        _ck(self.L.dlg_partition_rows(self.h, C.byref(n), C.byref(rows)), "partition_rows")
        return np.ctypeslib.as_array(rows, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.int32)

    def partition_stats(self):
        st = (C.c_long * len(PART_STAT_NAMES))()
        _ck(self.L.dlg_partition_stats(self.h, st, len(PART_STAT_NAMES)), "partition_stats")
        return {k: st[i] for i, k in enumerate(PART_STAT_NAMES)}

    def init_rccl(self, rank, nranks, unique_id):
        """join an RCCL communicator: collectives run on the backend's stream, no host in between"""
        buf = C.create_string_buffer(unique_id, 128)
        _ck(self.L.dlg_backend_init_rccl(self.h, rank, nranks, C.cast(buf, C.c_void_p)), "init_rccl")

    def set_noop_comm(self, on=True):
        """measurement only: one rank of a partition with every sum over the ranks skipped"""
        _ck(self.L.dlg_backend_set_noop_comm(self.h, 1 if on else 0), "set_noop_comm")

    def comm_size(self):
        n = C.c_int()
        _ck(self.L.dlg_backend_comm_size(self.h, C.byref(n)), "comm_size")
        return n.value

    def set_pattern(self, Jp, Ji):
        Jp = np.ascontiguousarray(Jp, dtype=np.int32)
        Ji = np.ascontiguousarray(Ji, dtype=np.int32)
        _ck(self.L.dlg_sparse_set_pattern(self.h, iptr(Jp), iptr(Ji)), "set_pattern")

    def stats(self):
        a, b = C.c_long(), C.c_long()
        c, d = C.c_int(), C.c_int()
        e = C.c_double()
        _ck(self.L.dlg_sparse_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)),
            "stats")
        return dict(nnz_JtJ_lower=a.value, nnz_L=b.value, n_supernodes=c.value, n_levels=d.value,
                    factor_flops=e.value)

    def schedule(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _ck(self.L.dlg_sparse_schedule(self.h, C.byref(a), C.byref(b), C.byref(c)), "schedule")
        return dict(n_levels=a.value, persist_level0=b.value, persist_items=c.value)

    def set_p(self, slot, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        _ck(self.L.dlg_point_set_p(self.h, slot, dptr(p)), "set_p")

    def upload(self, slot, x, J):
        x = np.ascontiguousarray(x, dtype=np.float64)
        J = np.ascontiguousarray(J, dtype=np.float64)
        self._keep = [x, J]
        _ck(self.L.dlg_point_upload(self.h, slot, dptr(x), dptr(J)), "upload")

    def upload_products(self, slot, norm2x, Jtx, JtJ):
        Jtx = np.ascontiguousarray(Jtx, dtype=np.float64)
        JtJ = np.ascontiguousarray(JtJ, dtype=np.float64)
        self._keep = [Jtx, JtJ]
        _ck(self.L.dlg_point_upload_products(self.h, slot, norm2x, dptr(Jtx), dptr(JtJ)), "upload")

    def bind_device(self, slot, x_ptr, J_ptr):
        _ck(self.L.dlg_point_bind_device(self.h, slot, C.c_void_p(x_ptr), C.c_void_p(J_ptr)), "bind")

    def eval(self, slot):
        a, b = C.c_double(), C.c_double()
        _ck(self.L.dlg_point_eval(self.h, slot, C.byref(a), C.byref(b)), "eval")
        return a.value, b.value

    def cauchy(self, slot):
        a = C.c_double()
        _ck(self.L.dlg_cauchy(self.h, slot, C.byref(a)), "cauchy")
        return a.value

    def factorize(self, slot, lam=0.0):
        ok = C.c_int()
        _ck(self.L.dlg_factorize(self.h, slot, lam, C.byref(ok)), "factorize")
        return bool(ok.value)

    def solve_gn(self, slot):
        a = C.c_double()
        _ck(self.L.dlg_solve_gn(self.h, slot, C.byref(a)), "solve_gn")
        return a.value

    def gauss_newton(self, slot, lam=0.0):
        """factorise (raising lambda as the reference does) + solve: returns (lambda, |gn|^2)"""
        l, a = C.c_double(lam), C.c_double()
        _ck(self.L.dlg_gauss_newton(self.h, slot, C.byref(l), C.byref(a)), "gauss_newton")
        return l.value, a.value

    def cauchy_gauss_newton(self, slot, lam=0.0):
        """Cauchy step + gauss_newton behind one synchronisation: (lambda, |cauchy|^2, |gn|^2)"""
        l, c, a = C.c_double(lam), C.c_double(), C.c_double()
        _ck(self.L.dlg_cauchy_gauss_newton(self.h, slot, C.byref(l), C.byref(c), C.byref(a)), "cauchy_gauss_newton")
        return l.value, c.value, a.value

    def _pnew_buffer(self):
        if self._pnew is None:
            self._pnew_ptr = self.L.dlg_host_alloc(8 * self.N)
            if not self._pnew_ptr:
                raise DlgError(self.L.dlg_last_error().decode())
            self._pnew = np.ctypeslib.as_array(C.cast(self._pnew_ptr, C.POINTER(C.c_double)), shape=(self.N,))
        return self._pnew

    def step(self, frm, to, kind, trustregion, want_p=True, tail=True):
        """make_step + expected_improvement behind one synchronisation:
        (|step|^2, k, max|step|, expected improvement, p_new); p_new as in make_step"""
        n2, k, am, ei = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        if want_p:
            self._pnew_buffer()
        _ck(self.L.dlg_step(self.h, frm, to, kind, trustregion, C.byref(n2), C.byref(k), C.byref(am),
                            C.byref(ei), dptr(self._pnew) if want_p else None), "step")
        e = ei.value
        if tail and self.step_tail_pending():
            e = self.step_tail()                # set_defer_tail: the value (and a page-locked p_new) complete here
        return n2.value, k.value, am.value, e, (self._pnew if want_p else None)

    def take_step(self, frm, to, trustregion, lam=0.0, want_p=True, tail=True):
        """Cauchy + Gauss-Newton + the choice of step + step + expected improvement behind one
        synchronisation: (lambda, dict(n2c, n2g, kind, n2s, k, amax, ei), p_new)"""
        l = C.c_double(lam)
        out = (C.c_double * 7)()
        if want_p:
            self._pnew_buffer()
        _ck(self.L.dlg_take_step(self.h, frm, to, trustregion, C.byref(l), out,
                                 dptr(self._pnew) if want_p else None), "take_step")
        keys = ("n2c", "n2g", "kind", "n2s", "k", "amax", "ei")
        r = dict(zip(keys, [float(v) for v in out]))
        r["kind"] = int(r["kind"])
        if tail and self.step_tail_pending():
            r["ei"] = self.step_tail()          # set_defer_tail: the value (and a page-locked p_new) complete here
        return l.value, r, (self._pnew if want_p else None)

    def run_steps(self, frm, to, nsteps, x_ptrs, J_ptrs, first_copy, trustregion, lam0=0.0):
        """nsteps x (bind the next resident copy, eval, take_step) in one C call (dlg_run_steps): returns
        (dict of the last step's scalars, kind of step)"""
        n = len(x_ptrs)
        xa = (C.c_void_p * n)(*x_ptrs)
        ja = (C.c_void_p * n)(*J_ptrs)
        out = (C.c_double * 9)()
        kind = C.c_int()
        _ck(self.L.dlg_run_steps(self.h, frm, to, nsteps, n, xa, ja, first_copy, trustregion, lam0, out, C.byref(kind)),
            "run_steps")
        keys = ("n2x", "n2c", "n2g", "k", "n2s", "ei", "gmax", "amax", "lam")
        return dict(zip(keys, [float(v) for v in out])), kind.value

    def solve_with_factor(self, slot, rhs):
        """(JtJ + lambda I) u = rhs with the factor held for `slot`; rhs: (N,) or (nrhs, N) rows"""
        r = np.ascontiguousarray(np.atleast_2d(rhs), dtype=np.float64)
        out = np.zeros_like(r)
        _ck(self.L.dlg_solve_with_factor(self.h, slot, dptr(r), dptr(out), r.shape[0]), "solve_with_factor")
        return out if np.ndim(rhs) == 2 else out[0]

    def solve_multi(self, slot, rhs):
        """blocked variant of solve_with_factor: rhs (nrhs, N) rows = right-hand sides; 16 per pass over the factor"""
        r = np.ascontiguousarray(np.atleast_2d(rhs), dtype=np.float64)
        out = np.zeros_like(r)
        _ck(self.L.dlg_solve_multi(self.h, slot, dptr(r), dptr(out), r.shape[0]), "solve_multi")
        return out

    def pseudoinverse_chunk(self, slot, row0, row1):
        """inv(JtJ + lambda I) Jt[:, row0:row1] as (row1 - row0, N): row i = the column of measurement row0 + i"""
        out = np.zeros((row1 - row0, self.N))
        _ck(self.L.dlg_pseudoinverse_chunk(self.h, slot, row0, row1, dptr(out)), "pseudoinverse_chunk")
        return out

    def make_step(self, frm, to, kind, trustregion, want_p=True):
        """p_new comes back in a page-locked buffer owned by this object (as the driver's operating
        points are): it is overwritten by the next call -- copy it to keep it."""
        n2, k, am = C.c_double(), C.c_double(), C.c_double()
        if want_p and self._pnew is None:
            self._pnew_ptr = self.L.dlg_host_alloc(8 * self.N)
            if not self._pnew_ptr:
                raise DlgError(self.L.dlg_last_error().decode())
            self._pnew = np.ctypeslib.as_array(C.cast(self._pnew_ptr, C.POINTER(C.c_double)), shape=(self.N,))
        _ck(self.L.dlg_make_step(self.h, frm, to, kind, trustregion, C.byref(n2), C.byref(k),
                                 C.byref(am), dptr(self._pnew) if want_p else None), "make_step")
        return n2.value, k.value, am.value, (self._pnew if want_p else None)

    def expected_improvement(self, frm, to):
        a = C.c_double()
        _ck(self.L.dlg_expected_improvement(self.h, frm, to, C.byref(a)), "expected_improvement")
        return a.value

    def download(self, slot, which, n=None):
        if n is None:
            n = self.M if which == VEC_X else self.N
        out = np.zeros(n)
        _ck(self.L.dlg_point_download(self.h, slot, which, dptr(out), n), "download")
        return out

    def factor_dense(self, n):
        out = np.zeros(n)
        _ck(self.L.dlg_factor_download_dense(self.h, dptr(out), n), "factor download")
        return out


def last_solve_timing():
    """{phase: (ms, calls)} of the calling thread's last dogleg_optimize* solve (DOGLEG_AMD_TIMING=1 must have been set)"""
    ms, n = (C.c_double * 7)(), (C.c_int * 7)()
    lib().dogleg_amd_last_solve_timing(ms, n)
    keys = ("pattern", "callback", "inputs", "point_eval", "take_step", "trace", "run_optimizer")
    return {k: (ms[i], n[i]) for i, k in enumerate(keys)}
