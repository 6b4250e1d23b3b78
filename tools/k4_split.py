#!/usr/bin/env python3
"""tools only: the assembly kernel's time with only the tasks of one shape running (DLG_ASM_ONLY_SHAPE=k; the
numbers it leaves are wrong, only the evaluation is called): how the 118 us of config #4 split between the
tasks of the point columns and those of the camera columns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdogleg_amd import capi
import problems as pb
if os.environ.get("DLG_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["DLG_LIB"])
prob = pb.BAProblem(2499, 45000, 500000, seed=11)
Jp, Ji = prob.pattern()
p = prob.p0()
x, Jx = prob.eval(p)
be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
be.set_pattern(Jp, Ji)
be.set_speculation(True)
be.set_p(0, p)
d = [(capi.DeviceArray(x), capi.DeviceArray(Jx)) for _ in range(3)]
for i in range(20):
    be.bind_device(0, d[i % 3][0].ptr, d[i % 3][1].ptr); be.eval(0)
be.set_profiling(True, only=["K4_kernel"])
for i in range(60):
    be.bind_device(0, d[i % 3][0].ptr, d[i % 3][1].ptr); be.eval(0)
ms, n = be.profile()["K4_kernel"]
print("DLG_ASM_ONLY_SHAPE =", os.environ.get("DLG_ASM_ONLY_SHAPE", "(all)"), ": k_assemble_mfma %.1f us per launch" % (1e3*ms/max(n, 1)))
be.close()
