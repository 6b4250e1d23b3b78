#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04c; mkdir -p $out
for wl in sparse-1m sparse-200k dense-50k; do
  timeout 600 python3 bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
done
timeout 600 python3 bench.py --workload sparse-5m --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_sparse-5m.json 2> $out/bench_sparse-5m.err
for f in $out/bench_*.json; do python3 tools/pj.py < $f; done
