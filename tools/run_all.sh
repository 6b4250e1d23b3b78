#!/bin/bash
# GPU test suite + a few bench lines (scratch runner for gpurun)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_ms_per_step']['K5_factor'], d['phases_ms_per_step']['K6_solve'])"; done
