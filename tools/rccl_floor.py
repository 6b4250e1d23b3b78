#!/usr/bin/env python3
"""tools only: what an RCCL all-reduce costs the library's stream at WORLD SIZE 1 (a gpurun box has one GPU): the floor of
the per-collective cost the multi-GPU projection has to assume -- launch + RCCL's own kernel for one rank, no wire.
Enqueue-to-completion, microseconds, back to back on the backend's stream (dlg_backend_time_allreduce).
usage: python tools/rccl_floor.py > profiles/rNN_rccl_floor.json      (run in a process of its own)"""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdogleg_amd import capi

N = 150006
be = capi.Backend(capi.DLG_DENSE, 64, 128, 0)
be.init_rccl(0, 1, capi.rccl_unique_id())
L = capi.lib()
L.dlg_backend_time_allreduce.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
out = {"what": "ncclAllReduce(double, sum) on the library's stream, world size 1, enqueue to completion, microseconds each (200 back to back)",
       "ranks": be.comm_size()}
for name, count in (("1_double_8B", 1), ("N_plus_1_doubles_1.2MB", N + 1), ("cut_buffer_1.18MB", 147500), ("64KB", 8192)):
    us = C.c_double()
    rc = L.dlg_backend_time_allreduce(be.h, count, 200, C.byref(us))
    assert rc == 0, L.dlg_last_error()
    out[name] = round(us.value, 2)
be.close()
print(json.dumps(out))
