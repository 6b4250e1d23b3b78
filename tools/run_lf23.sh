#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lf23; mkdir -p $out
timeout 900 python3 -m pytest tests/test_leaf_front_gpu.py -m gpu -x -q > $out/tests_lf.log 2>&1; echo "rc=$?" >> $out/tests_lf.log; tail -4 $out/tests_lf.log
DOGLEG_AMD_LEAF_FRONT=1 timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/tests_on.log 2>&1; echo "rc=$?" >> $out/tests_on.log; tail -4 $out/tests_on.log
DOGLEG_AMD_LEAF_FRONT=1 timeout 300 python3 bench.py --workload sparse-5m --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_lf_5m.json 2> $out/bench_lf_5m.err; python3 tools/pj.py < $out/bench_lf_5m.json
