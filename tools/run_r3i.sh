#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3i; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_sparse_gpu.py tests/test_scale_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -4 $out/tests.log
for wl in sparse-1m sparse-200k; do timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_$wl.json 2> $out/bench_$wl.err; python3 tools/pj.py < $out/bench_$wl.json; done
bash tools/run_prof.sh r3i/prof env
