#!/bin/bash
# tools only: panel_factor_b16: micro numbers, the GPU suite, bench lines
cd "$(dirname "$0")/.."
O=gpurun_out/b16; rm -rf $O; mkdir -p $O
for a in "100 66" "199 66" "127 126" "128 64"; do timeout 30 tools/micro/bench_panel 1 $a 512 | grep -E "B16"; done
timeout 30 tools/micro/bench_panel 1 128 64 | grep -E "MFMA |B16"
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
for i in 1 2; do
timeout 300 python3 bench.py --no-cpu-baseline | python3 tools/pj.py
done
timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-200k | python3 tools/pj.py
timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-5m --steps 30 | python3 tools/pj.py
timeout 300 python3 bench.py --no-cpu-baseline --workload dense-50k | python3 tools/pj.py
timeout 300 python3 bench.py --no-cpu-baseline --workload dense-50k > $O/dense.json; python3 -c "
import json; d=json.load(open('$O/dense.json')); print({k: round(v,4) for k,v in d['phases_ms_per_step'].items()})"
