#!/usr/bin/env python3
"""tools only: the dense JtJ kernel's time (HIP events around k_syrk_lower), config #2; DLG_LIB selects a variant
of the library (tools/variant_lib.sh) -- the factorisation may fail on a variant's wrong numbers, the kernel still runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdogleg_amd import capi
import problems as pb
if os.environ.get("DLG_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["DLG_LIB"])
prob = pb.DenseProblem(M=50000, N=2000, seed=11)
p = prob.p0()
x, J = prob.eval(p)
be = capi.Backend(capi.DLG_DENSE, prob.N, prob.M)
be.set_p(0, p)
dx, dJ = capi.DeviceArray(x), capi.DeviceArray(J)
be.bind_device(0, dx.ptr, dJ.ptr); be.eval(0)
for i in range(3):
    try: be.factorize(0, 1e-6)
    except Exception as e: print("factorize:", e)
be.set_profiling(True, only=["K4_kernel"])
for i in range(10):
    try: be.factorize(0, 1e-6 + 1e-9*i)
    except Exception as e: pass
ms, n = be.profile()["K4_kernel"]
print(os.environ.get("DLG_LIB", "(library)"), ": k_syrk_lower %.3f ms per launch (%d launches)" % (ms/max(n, 1), n))
be.close()
