#!/bin/bash
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${1:-check}; mkdir -p $out
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -6 $out/tests.log
for wl in sparse-1m sparse-200k; do
for i in 1 2; do
timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_${wl}_$i.json 2> $out/bench_${wl}_$i.err; python3 tools/pj.py < $out/bench_${wl}_$i.json
done
done
timeout 300 python3 bench.py --workload sparse-5m --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_5m.json 2> $out/bench_5m.err; python3 tools/pj.py < $out/bench_5m.json
timeout 300 python3 bench.py --workload dense-50k --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_dense.json 2> $out/bench_dense.err; python3 tools/pj.py < $out/bench_dense.json
