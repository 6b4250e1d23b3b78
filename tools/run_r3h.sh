#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3h; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_sparse_gpu.py tests/test_scale_gpu.py tests/test_sparse_patterns_gpu.py tests/test_shard_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -5 $out/tests.log
python3 tools/k4_split.py; DLG_ASM_ONLY_SHAPE=0 python3 tools/k4_split.py; DLG_ASM_ONLY_SHAPE=1 python3 tools/k4_split.py
for wl in sparse-1m sparse-200k; do timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_$wl.json 2> $out/bench_$wl.err; python3 tools/pj.py < $out/bench_$wl.json; done
