#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3d; mkdir -p $out
echo "== 200k default"; DOGLEG_AMD_DEBUG_SYNC=1 timeout 300 python3 bench.py --workload sparse-200k --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep -v "^libdogleg_amd: \(before\|after\).*no error" | cut -c1-300 | tail -5
echo "== 200k only-if-needed"; DOGLEG_AMD_SLICE_ONLY_IF_NEEDED=1 DOGLEG_AMD_DEBUG_SYNC=1 timeout 300 python3 bench.py --workload sparse-200k --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep -v "^libdogleg_amd: \(before\|after\).*no error" | cut -c1-300 | tail -5
echo "== tiny default"; DOGLEG_AMD_DEBUG_SYNC=1 timeout 300 python3 bench.py --workload sparse-tiny --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep -v "^libdogleg_amd: \(before\|after\).*no error" | cut -c1-300 | tail -5
echo "== 1m fill256"; DOGLEG_AMD_FRONT_FILL=256 DOGLEG_AMD_DEBUG_SYNC=1 timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep -v "^libdogleg_amd: \(before\|after\).*no error" | cut -c1-300 | tail -5
echo "== 1m fill256 replicas 2"; DOGLEG_AMD_FRONT_REPLICAS=2 DOGLEG_AMD_FRONT_FILL=256 DOGLEG_AMD_DEBUG_SYNC=1 timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep -v "^libdogleg_amd: \(before\|after\).*no error" | cut -c1-300 | tail -5
