#!/bin/bash
# round 6: the new expected-improvement tests, then value with / without the pass over J on one box
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r6a
timeout 1500 python3 -m pytest tests/test_expected_improvement_gpu.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r6a/ei_tests.txt
tail -15 gpurun_out/r6a/ei_tests.txt
timeout 1500 python3 -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "expected_improvement or retry or decision" 2>&1 | tail -5
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r6a/bench_a$i.json 2>gpurun_out/r6a/bench_a$i.err; python3 tools/pj.py < gpurun_out/r6a/bench_a$i.json
env DOGLEG_AMD_EI_JPASS=1 timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r6a/bench_b$i.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r6a/bench_b$i.json
env DOGLEG_AMD_NO_P_SIDE=1 timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r6a/bench_e$i.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r6a/bench_e$i.json
done
timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-200k > gpurun_out/r6a/bench_c.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r6a/bench_c.json
timeout 600 python3 bench.py --no-cpu-baseline --workload dense-50k --steps 20 > gpurun_out/r6a/bench_d.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r6a/bench_d.json
python3 - <<'PY'
import json
for n in ("a1","b1","e1"):
    d=json.loads([l for l in open(f"gpurun_out/r6a/bench_{n}.json") if l.startswith("{")][0])
    print(n, d["value"], d["phases_ms_per_step"], d["cached_retry_step"]["ms_per_step"], d["check"])
PY
