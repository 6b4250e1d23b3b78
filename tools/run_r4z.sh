#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for v in 256 512 1024 256 512; do
DOGLEG_AMD_PERSIST_MAX=$v timeout 600 python3 bench.py --workload sparse-5m --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 tools/pj.py
done
