#!/usr/bin/env python3
"""tools only: where a wave of the assembly kernel spends its clocks (a build with -DDLG_ASM_PROFILE: tools/variant_lib.sh build
sparse_assemble.hip -DDLG_ASM_PROFILE; DLG_LIB=tools/micro/libvar.so python tools/asm_prof.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdogleg_amd import capi
import problems as pb
capi.LIB_PATH = os.path.abspath(os.environ.get("DLG_LIB", "tools/micro/libvar.so"))
prob = pb.BAProblem(2499, 45000, 500000, seed=11)
Jp, Ji = prob.pattern()
p = prob.p0()
x, Jx = prob.eval(p)
be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
be.set_pattern(Jp, Ji)
be.set_speculation(True)
be.set_p(0, p)
d = [(capi.DeviceArray(x), capi.DeviceArray(Jx)) for _ in range(3)]
for i in range(6):
    be.bind_device(0, d[i % 3][0].ptr, d[i % 3][1].ptr); be.eval(0)
capi.lib().dlg_asm_profile_dump()
for i in range(30):
    be.bind_device(0, d[i % 3][0].ptr, d[i % 3][1].ptr); be.eval(0)
capi.lib().dlg_asm_profile_dump()
be.close()
