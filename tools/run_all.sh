#!/bin/bash
# GPU test suite + a few bench lines (scratch runner for gpurun)
cd "$(dirname "$0")/.." || exit 1
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for wl in sparse-1m sparse-1m sparse-200k sparse-5m; do timeout 300 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', round(d['value'],1), round(d['ms_per_step'],3), 'K5', round(d['phases_ms_per_step']['K5_factor'],3), 'K6', round(d['phases_ms_per_step']['K6_solve'],3), 'lv', d['symbolic']['n_levels'])"; done
