#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-soak}; mkdir -p $out
timeout 600 python3 tools/soak_steps.py 3000 sparse-1m > $out/soak_1m.log 2>&1; tail -2 $out/soak_1m.log
DOGLEG_AMD_LEAF_FRONT=1 timeout 600 python3 tools/soak_steps.py 3000 sparse-1m > $out/soak_lf_1m.log 2>&1; tail -2 $out/soak_lf_1m.log
DOGLEG_AMD_LEAF_FRONT=1 timeout 600 python3 tools/soak_steps.py 4000 sparse-200k > $out/soak_lf_200k.log 2>&1; tail -2 $out/soak_lf_200k.log
timeout 900 python3 tools/stress_patterns.py 120 9000 > $out/stress.log 2>&1; tail -2 $out/stress.log
DOGLEG_AMD_LEAF_FRONT=1 timeout 900 python3 tools/stress_patterns.py 120 9500 > $out/stress_lf.log 2>&1; tail -2 $out/stress_lf.log
