#!/usr/bin/env python3
"""tools only: leaf fronts (DOGLEG_AMD_LEAF_FRONT=1) on many bundle-adjustment-shaped problems of random sizes, every
form of the kernel at random, each checked against the oracle (tests/test_sparse_gpu.py::_ops_parity).
usage: stress_leaf_fronts.py [n] [seed0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdogleg_amd import capi
from tests import oracle_api as oa
from tests.test_sparse_gpu import _ops_parity

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
knobs = ["DOGLEG_AMD_LF_LISTS", "DOGLEG_AMD_LF_NO_RIDER", "DOGLEG_AMD_LF_NO_STRIDE", "DOGLEG_AMD_LF_NO_PF", "DOGLEG_AMD_LF_WGS"]
os.environ["DOGLEG_AMD_LEAF_FRONT"] = "1"
os.environ["DOGLEG_AMD_SYRK_MIN"] = "1"
bad = on = 0
for s in range(seed0, seed0 + n):
    rng = np.random.default_rng(s)
    for k in knobs:
        os.environ.pop(k, None)
    cfg = {}
    if rng.random() < 0.3: cfg["DOGLEG_AMD_LF_LISTS"] = "1"
    if rng.random() < 0.3: cfg["DOGLEG_AMD_LF_NO_RIDER"] = "1"
    if rng.random() < 0.3: cfg["DOGLEG_AMD_LF_NO_STRIDE"] = "1"
    if rng.random() < 0.3: cfg["DOGLEG_AMD_LF_NO_PF"] = "1"
    if rng.random() < 0.6: cfg["DOGLEG_AMD_LF_WGS"] = str(int(rng.choice([1, 2, 5, 16, 64])))
    os.environ.update(cfg)
    Nc = int(rng.integers(20, 120)); per = int(rng.integers(6, 22)); Np = Nc*per + int(rng.integers(0, Nc))
    Nobs = int(Np*rng.uniform(4.0, 14.0)); g = int(rng.choice([0, 2, 4, 6, 8]))
    prob = oa.BAProblem(Nc, Np, Nobs, g=g, seed=s)
    Jp, Ji = prob.pattern()
    lf = capi.symbolic_probe(prob.N, prob.M, Jp, Ji)["leaf_fronts"]
    on += lf
    try:
        st, err = _ops_parity(prob)
        print(f"seed {s}: Nc {Nc} Np {Np} Nobs {Nobs} g {g} leaf fronts {lf} {cfg} |gn diff| {err:.2e}", flush=True)
    except Exception as e:
        bad += 1
        print(f"seed {s}: Nc {Nc} Np {Np} Nobs {Nobs} g {g} leaf fronts {lf} {cfg} FAILED: {e}", flush=True)
print(f"{n - bad} of {n} passed, leaf fronts on in {on}")
