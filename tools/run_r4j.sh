#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for ns in 8 10 11 12 13 15 16 20; do echo "ns=$ns"; DLG_SYRK_NS=$ns timeout 600 python3 bench.py --workload dense-50k --no-cpu-baseline --steps 30 2>/dev/null | python3 tools/pj.py; done
