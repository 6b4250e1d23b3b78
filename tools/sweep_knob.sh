#!/bin/bash
# A/B sweep of one environment knob on one box: tools/sweep_knob.sh KNOB v1 v2 ...  (first the default, then each value; two rounds)
cd "$(dirname "$0")/.." || exit 1
K=$1; shift
for r in 1 2; do
  echo "default:"; timeout 600 python3 bench.py --no-cpu-baseline --steps 200 2>/dev/null | python3 tools/pj.py
  for v in "$@"; do
    echo "$K=$v:"; env $K=$v timeout 600 python3 bench.py --no-cpu-baseline --steps 200 2>/dev/null | python3 tools/pj.py
  done
done
