#!/bin/bash
# round 4, first GPU call: the new tests, then the baseline numbers of the code as it stands
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4a
timeout 1500 python3 -m pytest tests/test_rccl_entry_gpu.py tests/test_context_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4a/t1.log
timeout 1500 python3 -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "ahead_of_the_decision" 2>&1 | tail -15 > gpurun_out/r4a/t2.log
timeout 2400 python3 -m pytest tests/test_scale_gpu.py -x -q -m gpu -s -k "run_steps or summed_in_hbm" 2>&1 | tail -25 > gpurun_out/r4a/t3.log
timeout 600 python3 bench.py > gpurun_out/r4a/bench_sparse-1m.json 2> gpurun_out/r4a/bench_sparse-1m.err
timeout 600 python3 bench.py --workload sparse-5m --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4a/bench_sparse-5m.json 2> gpurun_out/r4a/bench_sparse-5m.err
cat gpurun_out/r4a/t1.log gpurun_out/r4a/t2.log gpurun_out/r4a/t3.log
python3 tools/pj.py < gpurun_out/r4a/bench_sparse-1m.json
