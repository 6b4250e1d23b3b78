#!/bin/bash
# tools only: the backward solve after a change: the GPU suite, the backward region's per-workgroup clocks, bench lines
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/k6; rm -rf $O; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 300 bash tools/prof_factor.sh run > $O/levels.txt 2>&1; grep -E "bwd wg +[0-9]:" $O/levels.txt | head -4
for i in 1 2; do timeout 300 python3 bench.py --no-cpu-baseline | python3 tools/pj.py; done
timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-200k | python3 tools/pj.py
timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-5m --steps 30 | python3 tools/pj.py
