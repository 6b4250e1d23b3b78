#!/bin/bash
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${1:-lf18}; mkdir -p $out
for a in "tiny" "tiny spec" "200k spec"; do
  echo "== $a"; timeout 300 python3 tools/lf_check.py $a 2>&1 | grep -v "^level" | tail -4 | head -3
  echo "== $a, 8 workgroups"; DOGLEG_AMD_LF_WGS=8 timeout 300 python3 tools/lf_check.py $a 2>&1 | grep -v "^level" | tail -4 | head -3
done > $out/check.log 2>&1
cat $out/check.log
DOGLEG_AMD_LEAF_FRONT=1 bash tools/prof_leaf.sh run > $out/prof.txt 2>&1
tail -1 $out/prof.txt
for wl in sparse-1m sparse-200k; do
DOGLEG_AMD_LEAF_FRONT=1 timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_$wl.json 2> $out/bench_$wl.err; python3 tools/pj.py < $out/bench_$wl.json
DOGLEG_AMD_LEAF_FRONT=1 DOGLEG_AMD_LF_NO_PF=1 timeout 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_nopf_$wl.json 2> $out/bench_nopf_$wl.err; python3 tools/pj.py < $out/bench_nopf_$wl.json
done
