#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4p; rm -rf $O; mkdir -p $O
for v in new base new base; do
  lib=libdogleg_amd/libdogleg_amd.so; [ $v = base ] && lib=tools/micro/libvar.so
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o t -- python3 tools/bench_lib.py $lib --steps 30 --warmup 5 > $O/b_$v.json 2>$O/e_$v.txt
  echo "== $v"
  f=$(find $O/p_$v -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('k_factor_level<256, true>','k_update_gather','k_update_fin')):
        print('   ', n[:60], r['Calls'], r['AverageNs'])
PY
  rm -rf $O/p_$v
  for i in 1 2; do timeout 300 python3 tools/bench_lib.py $lib 2>/dev/null | python3 tools/pj.py; done
done
