"""bench.py's contract (one JSON line on stdout; the keys the driver and the judge read), on the two tiny workloads."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("workload,dtype_bound", [("sparse-tiny", "hbm"), ("dense-tiny", "mfma")])
def test_bench_line_keeps_its_contract(gpu, workload, dtype_bound):
    d = _run(["--workload", workload, "--steps", "6", "--warmup", "2"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "steps/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["workload"] == workload and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] - 1e3) <= 1e-6 * 1e3
    assert d["untimed_presteps"] >= 30
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == dtype_bound and r["unit"] == ("GB/s" if dtype_bound == "hbm" else "TFLOP/s")
    assert r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
