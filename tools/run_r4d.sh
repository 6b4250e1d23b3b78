#!/bin/bash
# round 4: lazy expected improvement -- parity, whole suite, bench A/B, trace
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4d
timeout 900 python3 -m pytest tests/test_sparse_gpu.py -x -q -m gpu -s -k "lazy" 2>&1 | tail -12
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4d/bench_a$i.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4d/bench_a$i.json
DOGLEG_AMD_NO_LAZY_EI=1 timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4d/bench_b$i.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4d/bench_b$i.json
done
timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-200k > gpurun_out/r4d/bench_c.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4d/bench_c.json
timeout 600 python3 bench.py --workload sparse-5m --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4d/bench_e.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4d/bench_e.json
bash tools/run_trace.sh r4d/trace
