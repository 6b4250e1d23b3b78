#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['check'])"; done
timeout 300 python bench.py --no-cpu-baseline --workload sparse-5m --steps 4 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['check'])"
timeout 300 python bench.py --no-cpu-baseline --workload dense-50k 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['check'])"
