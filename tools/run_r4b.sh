#!/bin/bash
# round 4: the vector-pipe finish of a short last block (panel_finish_cols): micro numbers, parity, bench A/B
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4b
{
  for a in "199 66" "193 66" "127 54" "187 60" "100 66" "130 65" "131 67" "140 68" "127 126"; do echo "== nrows w = $a"; timeout 60 tools/micro/bench_panel 1 $a 512 | grep -E "B16"; done
} > gpurun_out/r4b/panel.txt 2>&1
cat gpurun_out/r4b/panel.txt
timeout 1500 python3 -m pytest tests/test_sparse_gpu.py tests/test_sparse_patterns_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4b/bench_a.json 2>/dev/null
DOGLEG_AMD_NO_VFIN=1 timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4b/bench_b.json 2>/dev/null
timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-200k > gpurun_out/r4b/bench_c.json 2>/dev/null
DOGLEG_AMD_NO_VFIN=1 timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-200k > gpurun_out/r4b/bench_d.json 2>/dev/null
for f in a b c d; do python3 tools/pj.py < gpurun_out/r4b/bench_$f.json; done
