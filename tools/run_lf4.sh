#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lf15; mkdir -p $out
for a in "tiny" "tiny spec" "200k spec"; do
  echo "== $a"; timeout 300 python3 tools/lf_check.py $a 2>&1 | grep -v "^level" | tail -6
done > $out/check.log 2>&1
cat $out/check.log
bash tools/prof_leaf.sh run > $out/prof.txt 2>&1
tail -3 $out/prof.txt
for wl in sparse-1m sparse-200k; do
timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_$wl.json 2> $out/bench_$wl.err; python3 tools/pj.py < $out/bench_$wl.json
done
bash tools/run_prof.sh lf15/prof env > /dev/null 2>&1; head -12 gpurun_out/lf15/prof/levels.txt | cut -c1-150
