"""Trial-by-trial comparison of two solver traces (SURVEY.md 8d 'Parity check')."""
import math
import numpy as np

STEP_TOL = 1e-10          # BASELINE.json north_star: per-iteration |dp| match within 1e-10


def compare_traces(a, b, step_tol=STEP_TOL, rel_scalar_tol=1e-8, what=("gpu", "oracle")):
    """a, b: libdogleg_amd.ctypes_defs.TraceBuffer.  Asserts same number of trials, same
    step type / accept decision / lambda on every trial and |step_a - step_b|_2 <= step_tol.
    Returns the max step difference."""
    assert a.ncallbacks == b.ncallbacks, f"callback count {a.ncallbacks} vs {b.ncallbacks}"
    assert a.ntrials == b.ntrials, f"trial count {a.ntrials} vs {b.ntrials}"
    ta, tb = a.trials(), b.trials()
    worst = 0.0
    for i, (x, y) in enumerate(zip(ta, tb)):
        for key in ("iteration", "accepted", "step_type", "did_step_to_edge"):
            assert x[key] == y[key], f"trial {i}: {key} {what[0]}={x[key]} {what[1]}={y[key]} (branch flip)"
        assert x["lambda_"] == y["lambda_"], f"trial {i}: lambda {x['lambda_']} vs {y['lambda_']}"
        d = float(np.linalg.norm(a.step[i] - b.step[i]))
        worst = max(worst, d)
        assert d <= step_tol, f"trial {i}: |step diff| = {d:.3e} > {step_tol:.1e}"
        dp = float(np.max(np.abs(a.p_trial[i] - b.p_trial[i])))
        assert dp <= step_tol * 10, f"trial {i}: |p_trial diff|_inf = {dp:.3e}"
        for key in ("norm2x_before", "norm2x_after", "norm2_step", "expected_improvement",
                    "trustregion_before", "trustregion_after"):
            u, v = x[key], y[key]
            if math.isnan(u) or math.isnan(v):
                assert math.isnan(u) and math.isnan(v), f"trial {i}: {key} {u} vs {v}"
                continue
            assert abs(u - v) <= rel_scalar_tol * max(1.0, abs(u), abs(v)), f"trial {i}: {key} {u} vs {v}"
    return worst
