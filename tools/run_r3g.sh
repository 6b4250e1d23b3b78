#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3g; mkdir -p $out
timeout 900 python3 -m pytest tests/test_device_eval_gpu.py tests/test_context_gpu.py tests/test_dropin_reference_programs.py tests/test_edge_cases_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -4 $out/tests.log
DOGLEG_AMD_TIMING=1 timeout 600 python3 tools/e2e_bench.py --workload sparse-1m > $out/e2e.json 2> $out/e2e_timing.err
cat $out/e2e.json; grep timing $out/e2e_timing.err | tail -8
