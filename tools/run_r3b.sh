#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3b; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_shard_gpu.py -m gpu -x -q > $out/tests_shard.log 2>&1; echo "rc=$?" >> $out/tests_shard.log
tail -15 $out/tests_shard.log
bash tools/run_prof.sh r3b/prof_rep1 env DOGLEG_AMD_FRONT_REPLICAS=1
bash tools/run_prof.sh r3b/prof_rep4 env DOGLEG_AMD_FRONT_REPLICAS=4 DOGLEG_AMD_FRONT_FILL=256
DOGLEG_AMD_TIMING=1 timeout 600 python3 tools/e2e_bench.py --workload sparse-1m > $out/e2e.json 2> $out/e2e_timing.err
cat $out/e2e.json
timeout 900 python3 -m pytest tests/test_edge_cases_gpu.py tests/test_scale_gpu.py -m gpu -x -q > $out/tests_scale.log 2>&1; echo "rc=$?" >> $out/tests_scale.log
tail -5 $out/tests_scale.log
