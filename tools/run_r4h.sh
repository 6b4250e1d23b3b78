#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4h
timeout 300 python3 tools/gpu_probe.py > gpurun_out/r4h/probe.txt 2>&1; cat gpurun_out/r4h/probe.txt
timeout 2400 python3 -m pytest tests/test_shard_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 1200 python3 tools/scaling_projection.py --workload sparse-1m --ranks 8 > gpurun_out/r4h/scaling_projection_sparse1m.md 2> gpurun_out/r4h/err1.txt
head -24 gpurun_out/r4h/scaling_projection_sparse1m.md; tail -3 gpurun_out/r4h/err1.txt
