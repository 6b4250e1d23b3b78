#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4i
timeout 1500 python3 -m pytest tests/test_dense_gpu.py tests/test_multi_rhs_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python3 -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "dense or config2" 2>&1 | tail -3
for i in 1 2; do timeout 600 python3 bench.py --workload dense-50k --no-cpu-baseline > gpurun_out/r4i/bench_dense_$i.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r4i/bench_dense_$i.json; done
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r4i/bench_dense_1.json'))
print({k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, d['roofline']['achieved'])
PY
