#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4g
timeout 300 python3 tools/gpu_probe.py > gpurun_out/r4g/probe.txt 2>&1; cat gpurun_out/r4g/probe.txt
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 600 python3 bench.py --no-cpu-baseline | python3 tools/pj.py
