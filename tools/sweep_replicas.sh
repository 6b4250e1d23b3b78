#!/bin/bash
# tools only: the replica count of the one-launch factor region against steps/s
cd "$(dirname "$0")/.." || exit 1
for r in 1 2 3 4 6 8; do echo "== DOGLEG_AMD_FRONT_REPLICAS=$r"; DOGLEG_AMD_FRONT_REPLICAS=$r timeout 300 python3 bench.py --no-cpu-baseline | python3 tools/pj.py; done
for f in 64 96 192; do echo "== DOGLEG_AMD_FRONT_FILL=$f"; DOGLEG_AMD_FRONT_FILL=$f timeout 300 python3 bench.py --no-cpu-baseline | python3 tools/pj.py; done
echo "== sparse-200k replicas"; for r in 2 4 8; do DOGLEG_AMD_FRONT_REPLICAS=$r timeout 300 python3 bench.py --no-cpu-baseline --workload sparse-200k | python3 tools/pj.py; done
