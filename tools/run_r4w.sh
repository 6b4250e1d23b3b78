#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 tools/pj.py
DOGLEG_AMD_MF_LEVEL=0 timeout 300 python3 bench.py --no-cpu-baseline 2>&1 | tail -3 | python3 tools/pj.py
DOGLEG_AMD_MF_LEVEL=0 DOGLEG_AMD_SYM_DEBUG=1 timeout 300 python3 bench.py --no-cpu-baseline --steps 5 2>&1 | grep -i "region\|factor level" | head -14
