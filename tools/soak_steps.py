#!/usr/bin/env python3
"""tools only: the same trust-region step of a config-#4-sized problem many thousand times (fresh
binding every time: nothing cached), alternating between two different inputs -- every repetition of
an input must reproduce its numbers bit for bit.  A stale read in one of the one-launch regions (a
workgroup reading an update matrix or x of an earlier launch) would show up as a different bit
pattern.  usage: soak_steps.py [steps] [sparse-1m|sparse-200k|dense-50k|dense-8k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdogleg_amd import capi
from tests import oracle_api as oa

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
workload = sys.argv[2] if len(sys.argv) > 2 else "sparse-1m"
if workload.startswith("dense"):
    # the one-launch dense factorisation and triangular solves (tile flags carry a per-launch epoch)
    prob = oa.DenseProblem({"dense-50k": 50000, "dense-8k": 8000}[workload], {"dense-50k": 2000, "dense-8k": 1000}[workload], seed=11)
    p = prob.p0()
    inputs = [prob.eval(p), prob.eval(p + 0.01)]
    be = capi.Backend(capi.DLG_DENSE, prob.N, prob.M)
else:
    shape = {"sparse-1m": (2499, 45000, 500000), "sparse-200k": (499, 9000, 100000)}[workload]
    prob = oa.BAProblem(*shape, seed=11, eps=0.4, p0_spread=0.6)
    Jp, Ji = prob.pattern()
    p = prob.p0()
    inputs = [prob.eval(p), prob.eval(p + 0.01)]
    be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
    be.set_pattern(Jp, Ji)
    be.set_speculation(True)
be.set_p(0, p)
# DLG_SOAK_DEFER=1: the expected improvement behind the decision point (dlg_backend_set_defer_tail), fetched at once;
# DLG_SOAK_DEFER=late: fetched behind the next evaluation, as the driver and dlg_run_steps do
defer = os.environ.get("DLG_SOAK_DEFER", "")
if defer:
    be.set_defer_tail(True)
pend = None
dev = [(capi.DeviceArray(np.ascontiguousarray(x)), capi.DeviceArray(np.ascontiguousarray(J))) for x, J in inputs]
ref, tr = [None, None], None
bad = 0
for k in range(steps):
    c = k & 1
    be.bind_device(0, dev[c][0].ptr, dev[c][1].ptr)
    n2x, gmax = be.eval(0)
    if pend is not None:
        pc, psig, pv = pend
        ei = be.step_tail()                     # (... and p_new of that step is complete now)
        psig = psig[:6] + (ei, float(pv[0]), float(pv[-1]), float(np.sum(pv)))
        pend = None
        if ref[pc] is None:
            ref[pc] = psig
        elif psig != ref[pc]:
            bad += 1
            if bad < 5:
                print("step", k - 1, "input", pc, "differs:", psig, "vs", ref[pc], flush=True)
    if tr is None:
        lam, n2c, n2g = be.cauchy_gauss_newton(0, 0.0)
        tr = 0.5*(n2c**0.5 + n2g**0.5)
        be.step(0, 1, capi.KIND_INTERP, tr)
        continue
    lam, r, pnew = be.take_step(0, 1, tr, 0.0, tail=(defer != "late"))
    sig = (n2x, gmax, r["n2c"], r["n2g"], r["n2s"], r["k"], r["ei"], float(pnew[0]), float(pnew[-1]), float(np.sum(pnew)))
    if defer == "late":
        pend = (c, sig, pnew)                   # (p_new of this step is complete behind the tail: looked at there)
        continue
    if ref[c] is None:
        ref[c] = sig
    elif sig != ref[c]:
        bad += 1
        if bad < 5:
            print("step", k, "input", c, "differs:", sig, "vs", ref[c], flush=True)
print(f"{steps} steps, {bad} deviations from the first result of their input")
sys.exit(1 if bad else 0)
