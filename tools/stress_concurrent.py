#!/usr/bin/env python3
"""tools only: several backends on ONE device at the same time (threads), each repeating the trust-region step of its
own problem -- far more workgroups of one-launch regions than the chip has CUs, so that workgroups start late and
out of step.  Every repetition of an input must reproduce its numbers bit for bit, and the first result of every
sparse thread must equal the result of the same problem run alone.  (The race between the replicas of a supernode,
profiles/r04_experiments.md, showed up only under this kind of sharing.)
usage: stress_concurrent.py [steps] [sparse threads] [dense threads]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdogleg_amd import capi
from tests import oracle_api as oa

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_sparse = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_dense = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bad, timeouts, sigs, lock = [], [], {}, threading.Lock()


def worker(name, kind, seed):
    try:
        if kind == "sparse":
            prob = oa.BAProblem(499, 9000, 100000, seed=seed, eps=0.4, p0_spread=0.6)
            Jp, Ji = prob.pattern()
            p = prob.p0()
            inputs = [prob.eval(p), prob.eval(p + 0.01)]
            be = capi.Backend(capi.DLG_SPARSE, prob.N, prob.M, prob.nnz)
            be.set_pattern(Jp, Ji)
            be.set_speculation(True)
        else:
            prob = oa.DenseProblem(8000, 1000, seed=seed)
            p = prob.p0()
            inputs = [prob.eval(p), prob.eval(p + 0.01)]
            be = capi.Backend(capi.DLG_DENSE, prob.N, prob.M)
        be.set_p(0, p)
        dev = [(capi.DeviceArray(np.ascontiguousarray(x)), capi.DeviceArray(np.ascontiguousarray(J))) for x, J in inputs]
        ref, tr, nbad = [None, None], None, 0
        for k in range(steps):
            c = k & 1
            be.bind_device(0, dev[c][0].ptr, dev[c][1].ptr)
            n2x, gmax = be.eval(0)
            if tr is None:
                lam, n2c, n2g = be.cauchy_gauss_newton(0, 0.0)
                tr = 0.5*(n2c**0.5 + n2g**0.5)
                be.step(0, 1, capi.KIND_INTERP, tr)
                continue
            lam, r, pnew = be.take_step(0, 1, tr, 0.0)
            sig = (n2x, gmax, r["n2c"], r["n2g"], r["n2s"], r["k"], r["ei"], float(pnew[0]), float(pnew[-1]), float(np.sum(pnew)))
            if ref[c] is None:
                ref[c] = sig
            elif sig != ref[c]:
                nbad += 1
                if nbad < 3:
                    print(name, "step", k, "input", c, "differs:", sig, "vs", ref[c], flush=True)
        be.close()
        with lock:
            sigs[name] = ref
            if nbad:
                bad.append((name, nbad))
    except Exception as e:
        with lock:
            # (a hand-off that timed out is REPORTED -- DLG_ERR_STATE, never a silently wrong number: on a chip this
            # crowded a workgroup that needs most of a CU's LDS may find no room for seconds; counted, tolerated)
            (timeouts if "timed out" in repr(e) else bad).append((name, repr(e)[:160]))


# alone first (the reference bits of sparse problem 0), then everybody at once
worker("alone", "sparse", 11)
alone = sigs["alone"]
th = [threading.Thread(target=worker, args=(f"sparse{i}", "sparse", 11)) for i in range(n_sparse)]
th += [threading.Thread(target=worker, args=(f"dense{i}", "dense", 3 + i)) for i in range(n_dense)]
[t.start() for t in th]
[t.join() for t in th]
for i in range(n_sparse):
    if f"sparse{i}" in sigs and sigs[f"sparse{i}"] != alone:
        bad.append((f"sparse{i}", "differs from the run alone"))
print(f"{n_sparse} sparse + {n_dense} dense backends at once, {steps} steps each: {len(bad)} silent deviations or other errors {bad[:4]}, "
      f"{len(timeouts)} reported hand-off time-outs {[t[0] for t in timeouts]}")
sys.exit(1 if bad else 0)
