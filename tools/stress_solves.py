#!/usr/bin/env python3
"""Random small nonlinear least-squares problems through the whole drop-in solver (dogleg_optimize2 /
dogleg_optimize_dense2, host callbacks) against the oracle, trial by trial: step kinds, accept
decisions, lambda, trust-region updates, steps within 1e-9.  usage: stress_solves.py [n] [seed0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_edge_cases_gpu import _dense_cb, _sparse_cb, _both
from tests import oracle_api as oa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
for s in range(seed0, seed0 + n):
    rng = np.random.default_rng(s)
    N = int(rng.integers(1, 13))
    M = int(N + rng.integers(0, 40))
    J0 = rng.standard_normal((M, N))
    xs = rng.standard_normal(M) * float(rng.choice([0.1, 1.0, 5.0]))
    nonlin = float(rng.choice([0.0, 0.3, 1.5, 3.0]))
    p0 = rng.standard_normal(N) * float(rng.choice([0.0, 1.0, 4.0]))
    prm = oa.default_params()
    prm.max_iterations = int(rng.integers(1, 30))
    prm.trustregion0 = float(rng.choice([1e-3, 1.0, 1e3, 1e6]))
    kind = "dense" if rng.random() < 0.5 else "sparse"
    cb = (_dense_cb if kind == "dense" else _sparse_cb)(J0, xs, M, N, nonlin)
    try:
        _both(kind, cb, p0, N, M, prm, tol=1e-9)
    except AssertionError as e:
        # Late in a solve the observed improvement is a difference of two norm2(x) that agree to the
        # last bits, so accept/reject decisions (and with them the trial count) are decided by
        # rounding -- in the reference too.  Such a divergence is benign when everything before the
        # noise floor agrees and both solves end at the same cost.
        import ctypes as C
        from libdogleg_amd import capi
        addr = C.cast(cb, C.c_void_p)
        nnz = M * N if kind == "sparse" else 0
        ro, po, tro = oa.oracle_solve(kind, p0, N, M, nnz, addr, None, prm)
        rg, pg, trg = capi.optimize(kind, p0, N, M, nnz, addr, None, prm)
        tg, to = trg.trials(), tro.trials()
        floor = next((i for i, t in enumerate(to) if abs(t["expected_improvement"]) <= 1e-10 * max(1.0, t["norm2x_before"])), len(to))
        same = all(tg[i]["accepted"] == to[i]["accepted"] and tg[i]["step_type"] == to[i]["step_type"] and
                   np.linalg.norm(trg.step[i] - tro.step[i]) <= 1e-8 for i in range(min(floor, len(tg), len(to))))
        if same and abs(rg - ro) <= 1e-9 * max(1.0, abs(ro)) and floor < max(len(tg), len(to)):
            print("ok-ish seed", s, f"diverges only below the noise floor (trial {floor} of {len(to)}); same final cost", flush=True)
            continue
        bad += 1
        print("FAIL seed", s, kind, "N", N, "M", M, "nonlin", nonlin, "tr0", prm.trustregion0, repr(e)[:240], flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed", s, kind, "N", N, "M", M, "nonlin", nonlin, "tr0", prm.trustregion0, repr(e)[:240], flush=True)
print(f"{n - bad} of {n} passed")
sys.exit(1 if bad else 0)
