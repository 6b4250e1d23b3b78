#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in none 128 64 192; do
echo "FAC512=$v"; if [ $v = none ]; then unset DLG_FAC512; else export DLG_FAC512=$v; fi
timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_ms_per_step']['K5_factor'])"
done
