#!/bin/bash
# round 4: A/B of one knob: tools/ab_knob.sh KNOB [pytest -k expr]
cd "$(dirname "$0")/.." || exit 1
K=$1; mkdir -p gpurun_out/r4e
timeout 1500 python3 -m pytest tests/test_sparse_gpu.py tests/test_sparse_patterns_gpu.py tests/test_scale_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4e/bench_a$i.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4e/bench_a$i.json
env $K=1 timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4e/bench_b$i.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4e/bench_b$i.json
done
timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-200k > gpurun_out/r4e/bench_c.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4e/bench_c.json
env $K=1 timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-200k > gpurun_out/r4e/bench_d.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4e/bench_d.json
timeout 600 python3 bench.py --workload sparse-5m --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4e/bench_e.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4e/bench_e.json
bash tools/run_trace.sh r4e/trace | grep -v "other queue"
