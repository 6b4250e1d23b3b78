#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate: dogleg_optimize2 with a host callback on a
synthetic block-arrowhead problem.  Every evaluation pays the callback on the
host plus the H2D of x and the Jacobian values; this is the number a drop-in
user sees, and it is NOT bench.py's `value` (inputs resident in HBM)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdogleg_amd import capi
from tests import oracle_api as oa

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sparse-1m")
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--oracle", action="store_true", help="also run the CPU oracle end to end and compare")
a = ap.parse_args()
cfg = {"sparse-1m": (2499, 45000, 500000), "sparse-200k": (499, 9000, 100000), "sparse-tiny": (49, 900, 10000)}[a.workload]
prob = oa.BAProblem(*cfg, seed=11, eps=0.4, p0_spread=0.6)
prm = oa.default_params()
prm.max_iterations = a.iters
prm.trustregion0 = 20.0
p0 = prob.p0()
t0 = time.perf_counter()
x, Jx = prob.eval(p0)
t_cb = time.perf_counter() - t0
t0 = time.perf_counter()
r, p, tr = capi.optimize("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
t_all = time.perf_counter() - t0
out = {"workload": a.workload, "Nmeas": prob.M, "Nstate": prob.N, "nnz": prob.nnz,
       "trials": tr.ntrials, "callbacks": tr.ncallbacks, "total_s": t_all,
       "callback_s_each": t_cb, "callbacks_s_total": t_cb * tr.ncallbacks,
       "end_to_end_steps_per_s": tr.ntrials / t_all,
       "steps_per_s_excluding_callback": tr.ntrials / max(1e-9, t_all - t_cb * tr.ncallbacks),
       "h2d_bytes_per_eval": 8 * (prob.nnz + prob.M), "norm2x": r,
       "step_types": [t["step_type"] for t in tr.trials()],
       "note": "total includes the one-off symbolic analysis and pinned allocations"}
if a.oracle:
    t0 = time.perf_counter()
    ro, po, tro = oa.oracle_solve("sparse", p0, prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    out["oracle_total_s"] = time.perf_counter() - t0
    out["max_abs_p_diff_vs_oracle"] = float(np.max(np.abs(p - po)))
    out["same_trial_count"] = tro.ntrials == tr.ntrials
print(json.dumps(out))
