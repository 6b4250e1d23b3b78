#!/bin/bash
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${1:-dense}; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_dense_gpu.py tests/test_edge_cases_gpu.py tests/test_dropin_reference_programs.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log; tail -4 $out/tests.log
timeout 300 python3 bench.py --workload dense-50k --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_dense.json 2> $out/bench_dense.err; python3 tools/pj.py < $out/bench_dense.json
timeout 300 python3 tools/soak_steps.py 1500 dense-8k > $out/soak_dense.log 2>&1; tail -1 $out/soak_dense.log
