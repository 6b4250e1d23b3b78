#!/bin/bash
cd "$(dirname "$0")/.."
for v in "" "DOGLEG_AMD_B16=1 DOGLEG_AMD_B16_MAXW=128" "DOGLEG_AMD_B16=1 DOGLEG_AMD_B16_MAXW=192" "DOGLEG_AMD_B16=1 DOGLEG_AMD_B16_MAXW=256" "DOGLEG_AMD_B16=1"; do
echo "== $v"
env $v timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-5m --steps 30 | python3 tools/pj.py
done
