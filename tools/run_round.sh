#!/bin/bash
# tools only: the new sweep tests, the micro numbers of the panel sweeps, then the round's artifacts
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r03
timeout 900 python3 -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "blocks_of_16" 2>&1 | tail -3
{
  echo "# tools/micro/bench_panel: one workgroup of 512 threads, panel resident in LDS; us per launch include the load / store of the panel (about 5 us)"
  echo "# MFMA = panel_factor_mfma (blocks of 8 columns, diagonal wave + barriers), B16 = panel_factor_b16 (blocks of 16, diagonal tile in registers)"
  for a in "92 60" "100 66" "199 66" "187 60" "127 126" "127 54" "128 64"; do echo "== nrows w = $a"; timeout 60 tools/micro/bench_panel 1 $a 512 | grep -E "MFMA |B16"; done
  echo "# latencies (tools/micro/bench_mfma_lat, clocks per dependent iteration)"
  timeout 30 tools/micro/bench_mfma_lat
} > gpurun_out/r03/panel_sweep.txt 2>&1
tail -12 gpurun_out/r03/panel_sweep.txt
bash tools/collect_round.sh r03 > gpurun_out/r03/collect.log 2>&1
python3 tools/pj.py < gpurun_out/r03/bench_sparse-1m.json; python3 tools/pj.py < gpurun_out/r03/bench_dense-50k.json
