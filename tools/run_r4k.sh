#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 1500 python3 -m pytest tests/test_dense_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "dense or config2" 2>&1 | tail -3
for i in 1 2; do
timeout 600 python3 bench.py --workload dense-50k --no-cpu-baseline --steps 50 2>/dev/null | python3 tools/pj.py
DOGLEG_AMD_SYRK_4WAVES=1 timeout 600 python3 bench.py --workload dense-50k --no-cpu-baseline --steps 50 2>/dev/null | python3 tools/pj.py
done
