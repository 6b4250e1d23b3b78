#!/bin/bash
# round 3, first GPU pass: the new tests, then the replica sweep on sparse-1m
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3a; mkdir -p $out
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
for cfg in "1 128" "4 128" "4 256" "2 256" "8 256"; do
  set -- $cfg
  DOGLEG_AMD_FRONT_REPLICAS=$1 DOGLEG_AMD_FRONT_FILL=$2 timeout 300 python3 bench.py --no-cpu-baseline --steps 60 --warmup 10 > $out/bench_rep$1_fill$2.json 2> $out/bench_rep$1_fill$2.err
done
DOGLEG_AMD_TIMING=1 timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $out/timing.err
timeout 300 python3 bench.py --workload sparse-200k --no-cpu-baseline > $out/bench_200k.json 2> $out/bench_200k.err
timeout 300 python3 bench.py --workload dense-50k --no-cpu-baseline > $out/bench_dense.json 2> $out/bench_dense.err
tail -5 $out/tests.log
for f in $out/bench_rep*.json $out/bench_200k.json $out/bench_dense.json; do python3 tools/pj.py < $f; done
