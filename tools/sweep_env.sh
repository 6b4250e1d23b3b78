#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   (one bench line per value; scratch helper for gpurun)
cd "$(dirname "$0")/.." || exit 1
var=$1; shift
for v in "$@"; do
  export $var=$v
  echo -n "$var=$v  "
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 3 $BENCH_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), 'K5', round(d['phases_ms_per_step']['K5_factor'],3), 'K6', round(d['phases_ms_per_step']['K6_solve'],3), 'lv', d['symbolic']['n_levels'], 'sn', d['symbolic']['n_supernodes'])"
done
