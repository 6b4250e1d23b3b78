#!/bin/bash
# tools only: the factor-and-solve-ahead path: its test, the take_step tests, bench lines with and without
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/presolve; O=gpurun_out/presolve
timeout 900 python3 -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "ahead or take_step or speculative or fused" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for i in 1 2; do
timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_on_$i.json 2>$O/err.txt; cat $O/bench_on_$i.json | cut -c1-200
DOGLEG_AMD_NO_PRESOLVE=1 timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_off_$i.json 2>>$O/err.txt; cat $O/bench_off_$i.json | cut -c1-200
done
DOGLEG_AMD_LEAF_FRONT=1 timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_lf.json 2>>$O/err.txt; cat $O/bench_lf.json | cut -c1-200
timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-200k > $O/bench_200k.json 2>>$O/err.txt; cat $O/bench_200k.json | cut -c1-200
