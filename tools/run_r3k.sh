#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3k; mkdir -p $out
for wl in sparse-1m sparse-200k dense-50k; do timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_$wl.json 2> $out/bench_$wl.err; python3 tools/pj.py < $out/bench_$wl.json; tail -3 $out/bench_$wl.err; done
timeout 300 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; python3 tools/pj.py < $out/bench_default.json; tail -3 $out/bench_default.err
timeout 600 python3 -m pytest tests/test_sparse_gpu.py -m gpu -x -q -k "leak or returned" 2>&1 | tail -3
python3 tools/host_gap.py
