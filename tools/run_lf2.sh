#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lf14; mkdir -p $out
for wl in sparse-1m sparse-200k; do
timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_$wl.json 2> $out/bench_$wl.err; python3 tools/pj.py < $out/bench_$wl.json
DOGLEG_AMD_NO_LEAF_FRONT=1 timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_off_$wl.json 2> $out/bench_off_$wl.err; python3 tools/pj.py < $out/bench_off_$wl.json
done
timeout 300 python3 bench.py --workload sparse-5m --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_5m.json 2> $out/bench_5m.err; python3 tools/pj.py < $out/bench_5m.json
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -15 $out/tests.log
DOGLEG_AMD_SYRK_MIN=1 timeout 1500 python3 -m pytest tests/test_sparse_gpu.py tests/test_scale_gpu.py tests/test_edge_cases_gpu.py -m gpu -q > $out/tests_min1.log 2>&1; echo "rc=$?" >> $out/tests_min1.log
tail -15 $out/tests_min1.log
