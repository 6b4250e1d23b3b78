#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ that are produced HERE (the
sample_trace.json fixture is transcribed from SURVEY.md Appendix B instead):

  sample_measurements.json  the 100 simulated measurements + start point of the
                            reference's sample problem: glibc srandom(0)/random()
                            stream (sample.c:46-62, 351, 370-371), produced by
                            problems/problems.c:sample_init()
  oracle_ba_tiny.json       per-trial trace of the CPU oracle on a tiny synthetic
                            block-arrowhead problem (pins the GPU path against a
                            committed vector, not only against a live oracle run)
  oracle_dense_small.json   same for a small dense problem
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import oracle_api as oa
from libdogleg_amd.ctypes_defs import dptr

HERE = os.path.dirname(os.path.abspath(__file__))


def hexlist(a):
    return [float(v).hex() for v in np.asarray(a).ravel()]


def trace_json(tr):
    out = []
    for i, t in enumerate(tr.trials()):
        d = {k: (None if isinstance(v, float) and v != v else v) for k, v in t.items()}
        d["step_hex"] = hexlist(tr.step[i])
        d["p_trial_hex"] = hexlist(tr.p_trial[i])
        out.append(d)
    return out


def main():
    P = oa.problems()
    p0 = np.zeros(6)
    P.sample_init(dptr(p0))
    m = np.zeros(100)
    P.sample_get_measurements(dptr(m))
    json.dump({"_generator": "tests/golden/make_goldens.py (glibc random() after srandom(0))",
               "measurements_hex": hexlist(m), "p0_hex": hexlist(p0)},
              open(os.path.join(HERE, "sample_measurements.json"), "w"), indent=0)

    prob = oa.BAProblem(4, 20, 60, seed=2, eps=0.4, p0_spread=0.8)
    prm = oa.default_params()
    prm.max_iterations = 15
    prm.trustregion0 = 1.0
    r, p, tr = oa.oracle_solve("sparse", prob.p0(), prob.N, prob.M, prob.nnz, prob.cb, prob.cookie, prm)
    json.dump({"_generator": "tests/golden/make_goldens.py: oracle on BAProblem(4,20,60,seed=2,eps=0.4,p0_spread=0.8), "
                             "max_iterations=15, trustregion0=1.0",
               "norm2x": float(r).hex(), "p_final_hex": hexlist(p), "ncallbacks": tr.ncallbacks,
               "trials": trace_json(tr)},
              open(os.path.join(HERE, "oracle_ba_tiny.json"), "w"), indent=0)

    dp = oa.DenseProblem(M=300, N=24, seed=9, eps=0.4, p0_spread=0.8)
    prm = oa.default_params()
    prm.max_iterations = 12
    prm.trustregion0 = 0.5
    r, p, tr = oa.oracle_solve("dense", dp.p0(), dp.N, dp.M, 0, dp.cb, dp.cookie, prm)
    json.dump({"_generator": "tests/golden/make_goldens.py: oracle on DenseProblem(300,24,seed=9,eps=0.4,p0_spread=0.8), "
                             "max_iterations=12, trustregion0=0.5",
               "norm2x": float(r).hex(), "p_final_hex": hexlist(p), "ncallbacks": tr.ncallbacks,
               "trials": trace_json(tr)},
              open(os.path.join(HERE, "oracle_dense_small.json"), "w"), indent=0)
    print("goldens written")


if __name__ == "__main__":
    main()
