#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4f
timeout 1200 python3 tools/scaling_projection.py --workload sparse-1m --ranks 8 > gpurun_out/r4f/scaling_projection_sparse1m.md 2> gpurun_out/r4f/err1.txt
cat gpurun_out/r4f/scaling_projection_sparse1m.md; tail -3 gpurun_out/r4f/err1.txt
