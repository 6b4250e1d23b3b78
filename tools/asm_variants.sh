#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for g in 64 256 32 8 1; do
echo "SYRK_MIN=$g"; DOGLEG_AMD_SYRK_MIN=$g timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_ms_per_step']['K5_factor'], d['phases_ms_per_step']['K6_solve'])"
done
