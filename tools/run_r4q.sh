#!/bin/bash
cd "$(dirname "$0")/.." || exit 1

for v in new old new old; do
  lib=tools/micro/libk4_$v.so; [ $v = new ] && lib=libdogleg_amd/libdogleg_amd.so
  echo "== $v"
  DLG_LIB=$lib timeout 200 python3 tools/k4_split.py 2>/dev/null | tail -1
  DLG_LIB=$lib DLG_ASM_ONLY_SHAPE=1 timeout 200 python3 tools/k4_split.py 2>/dev/null | tail -1
  timeout 300 python3 tools/bench_lib.py $lib 2>/dev/null | python3 tools/pj.py
done
