#!/bin/bash
cd $GRAFT_REPO_ROOT

for cc in 8192 12288 16384; do
echo "CHAIN_CAP=$cc"; DOGLEG_AMD_CHAIN_CAP=$cc timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_ms_per_step']['K5_factor'], d['phases_ms_per_step']['K6_solve'], d['symbolic']['n_levels'], d['symbolic']['n_supernodes'])"
done
