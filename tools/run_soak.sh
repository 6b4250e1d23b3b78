#!/bin/bash
# tools only: long runs at the current defaults (block-16 sweep, step prepared at evaluation time) and with the opt-ins
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${1:-soak2}; mkdir -p $out
timeout 600 python3 tools/soak_steps.py 3000 sparse-1m > $out/soak_1m.log 2>&1; tail -2 $out/soak_1m.log
timeout 600 python3 tools/soak_steps.py 4000 sparse-200k > $out/soak_200k.log 2>&1; tail -2 $out/soak_200k.log
timeout 900 python3 tools/stress_patterns.py 120 9000 > $out/stress.log 2>&1; tail -2 $out/stress.log
timeout 900 python3 tools/stress_dense.py > $out/stress_dense.log 2>&1; tail -2 $out/stress_dense.log
timeout 900 python3 tools/stress_solves.py > $out/stress_solves.log 2>&1; tail -2 $out/stress_solves.log
