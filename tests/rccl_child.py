"""Child process of tests/test_rccl_entry_gpu.py: a full solve through the PUBLIC multi-GPU entry of the library
with RCCL as the communicator, at world size 1 (a gpurun box has one GPU; RCCL refuses two ranks on one device),
compared trial by trial with the oracle's trace.  Run in a process of its own so that a communicator that does
not come up costs the test its time-out and not the suite.

usage: python -m tests.rccl_child api|env  sparse|dense|device
  api: dogleg_amd_set_communicator(0, 1, 0, id) on the calling thread, then dogleg_optimize2 / _dense2 / _device2
  env: nothing is called -- the parent set DOGLEG_AMD_WORLD_SIZE=1 DOGLEG_AMD_FORCE_COMM=1 DOGLEG_AMD_RANK=0
       DOGLEG_AMD_RCCL_ID_FILE=<path> (rank 0 writes the id file, the communicator is the process's)
Prints "OK <ntrials> <worst step difference> <ranks RCCL reports>" on success."""
import ctypes as C
import sys
import numpy as np

from libdogleg_amd import capi
from tests import oracle_api as oa
from tests.parity import compare_traces


def main():
    how, kind = sys.argv[1], sys.argv[2]
    L = capi.lib()
    if how == "api":
        ident = (C.c_ubyte * 128)()
        assert L.dogleg_amd_rccl_unique_id(ident) == 0, L.dlg_last_error()
        assert L.dogleg_amd_set_communicator(0, 1, 0, ident) == 0
    prm = oa.default_params()
    twin = None
    if kind == "dense":
        prob = oa.DenseProblem(M=1201, N=96, seed=2)
        prm.max_iterations = 8
        nnz = 0
    else:
        prob = oa.BAProblem(49, 900, 10000, seed=4, eps=0.4, p0_spread=0.6)
        prm.max_iterations = 12
        prm.trustregion0 = 3.0
        nnz = prob.nnz
    okind = "dense" if kind == "dense" else "sparse"
    ro, po, tro = oa.oracle_solve(okind, prob.p0(), prob.N, prob.M, nnz, prob.cb, prob.cookie, prm)
    ctx = C.c_void_p()
    p = prob.p0().copy()
    tr = capi.TraceBuffer(prob.N, 256)
    L.dlg_set_trace(C.cast(tr.byref(), C.c_void_p))
    if kind == "device":
        twin = oa.DeviceTwin(prob)
        Jp, Ji = prob.pattern()
        r = L.dogleg_optimize_device2(capi.dptr(p), prob.N, prob.M, nnz, capi.iptr(Jp), capi.iptr(Ji), twin.cb, twin.cookie,
                                      C.byref(prm), C.byref(ctx))
    elif kind == "sparse":
        r = L.dogleg_optimize2(capi.dptr(p), prob.N, prob.M, nnz, prob.cb, prob.cookie, C.byref(prm), C.byref(ctx))
    else:
        r = L.dogleg_optimize_dense2(capi.dptr(p), prob.N, prob.M, prob.cb, prob.cookie, C.byref(prm), C.byref(ctx))
    L.dlg_set_trace(None)
    assert r >= 0 and ctx.value, "the solve failed"
    # the solve really ran as one rank of a communicator, and that communicator is RCCL's
    nr = C.c_int(-1)
    assert L.dogleg_amd_rank(ctx, C.byref(nr)) == 0 and nr.value == 1
    be = L.dogleg_amd_backend(ctx)
    n = C.c_int(0)
    assert L.dlg_backend_comm_size(be, C.byref(n)) == 0
    assert L.dlg_backend_has_rccl(be) == 1, "the backend of the solve holds no RCCL communicator"
    w = compare_traces(tr, tro)
    assert np.max(np.abs(p - po)) <= 1e-10
    assert abs(r - ro) <= 1e-9 * max(1.0, ro)
    L.dogleg_freeContext(C.byref(ctx))
    if twin:
        twin.close()
    L.dogleg_amd_clear_communicator()
    print(f"OK {tr.ntrials} {w:.3e} {n.value}")


if __name__ == "__main__":
    main()
