#!/usr/bin/env python3
"""tools only: the assembly kernel's time in THIS process beside the device addresses of what it reads and writes
(its time has two modes, 0.094 / 0.102 ms on config #4, that change from process to process on one box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdogleg_amd import capi
import problems as pb
prob = pb.BAProblem(2499, 45000, 500000, seed=11)
Jp, Ji = prob.pattern()
p = prob.p0()
x, Jx = prob.eval(p)
pad = int(os.environ.get("K4_PAD", "0"))
junk = capi.DeviceArray(np.zeros(max(pad, 1))) if pad else None
be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
be.set_pattern(Jp, Ji)
be.set_speculation(True)
be.set_p(0, p)
d = [(capi.DeviceArray(x), capi.DeviceArray(Jx)) for _ in range(3)]
for i in range(20):
    be.bind_device(0, d[i % 3][0].ptr, d[i % 3][1].ptr); be.eval(0)
be.set_profiling(True, only=["K4_kernel"])
for i in range(90):
    be.bind_device(0, d[i % 3][0].ptr, d[i % 3][1].ptr); be.eval(0)
ms, n = be.profile()["K4_kernel"]
print("k_assemble_mfma %.1f us   J copies at %s   x at %s" % (1e3*ms/max(n, 1), " ".join(hex(a[1].ptr) for a in d), " ".join(hex(a[0].ptr) for a in d)))
be.close()
