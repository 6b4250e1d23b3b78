"""tools only: host time per backend call of one steady-state step of config #4 (bind / eval / take_step):
what the host adds to the kernels' span (rocprofv3 inflates it: measure without the profiler)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from libdogleg_amd import capi
from tests import oracle_api as oa
prob = oa.BAProblem(2499, 45000, 500000, seed=11)
Jp, Ji = prob.pattern(); p0 = prob.p0(); x, Jx = prob.eval(p0)
be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
be.set_pattern(Jp, Ji)
d = [(capi.DeviceArray(np.ascontiguousarray(x)), capi.DeviceArray(np.ascontiguousarray(Jx))) for _ in range(3)]
be.set_p(0, p0); be.set_speculation(True)
be.bind_device(0, d[0][0].ptr, d[0][1].ptr); be.eval(0)
lam, n2c, n2g = be.cauchy_gauss_newton(0, 0.0); tr = 0.5*(n2c**0.5 + n2g**0.5)
be.step(0, 1, capi.KIND_INTERP, tr)
pc = time.perf_counter
tb = te = tt = 0.0
n = 300
for i in range(20 + n):
    if i == 20: tb = te = tt = 0.0; t00 = pc()
    c = i % 3
    t0 = pc(); be.bind_device(0, d[c][0].ptr, d[c][1].ptr)
    t1 = pc(); be.eval(0)
    t2 = pc(); be.take_step(0, 1, tr, 0.0)
    t3 = pc(); tb += t1 - t0; te += t2 - t1; tt += t3 - t2
tot = pc() - t00
print(f"per step: bind {tb/n*1e6:.1f} us  eval {te/n*1e6:.1f} us  take_step {tt/n*1e6:.1f} us  sum {(tb+te+tt)/n*1e6:.1f}  loop {tot/n*1e6:.1f} us")
