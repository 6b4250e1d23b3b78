#!/bin/bash
# tools only: panel_factor_b16 in the factor kernel: the GPU suite, per-level table, bench on/off
cd "$(dirname "$0")/.."
O=gpurun_out/b16; rm -rf $O; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 300 bash tools/prof_factor.sh run > $O/levels.txt 2>&1; grep -E "^level" $O/levels.txt | cut -c1-260
for i in 1 2; do
timeout 300 python3 bench.py --no-cpu-baseline | python3 tools/pj.py
DOGLEG_AMD_NO_B16=1 timeout 300 python3 bench.py --no-cpu-baseline | python3 tools/pj.py
done
timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-200k | python3 tools/pj.py
timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-5m --steps 30 | python3 tools/pj.py
DOGLEG_AMD_NO_B16=1 timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-5m --steps 30 | python3 tools/pj.py
DOGLEG_AMD_LEAF_FRONT=1 timeout 300 python3 bench.py --no-cpu-baseline | python3 tools/pj.py
