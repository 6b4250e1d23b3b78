#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3c; mkdir -p $out
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
tail -6 $out/tests.log
for cfg in "4 128" "4 256" "8 256" "4 512"; do
  set -- $cfg
  DOGLEG_AMD_FRONT_REPLICAS=$1 DOGLEG_AMD_FRONT_FILL=$2 timeout 300 python3 bench.py --no-cpu-baseline --steps 60 --warmup 10 > $out/bench_rep$1_fill$2.json 2> $out/bench_rep$1_fill$2.err
  python3 tools/pj.py < $out/bench_rep$1_fill$2.json
done
bash tools/run_prof.sh r3c/prof_rep4 env DOGLEG_AMD_FRONT_REPLICAS=4 DOGLEG_AMD_FRONT_FILL=256
timeout 300 python3 bench.py --workload sparse-200k --no-cpu-baseline > $out/bench_200k.json 2> $out/bench_200k.err; python3 tools/pj.py < $out/bench_200k.json
timeout 300 python3 bench.py --workload sparse-5m --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_5m.json 2> $out/bench_5m.err; python3 tools/pj.py < $out/bench_5m.json
