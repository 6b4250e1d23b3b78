#!/bin/bash
# tools only: tools/ab_var.sh [workload] -- the library against tools/micro/libvar.so (tools/variant_lib.sh), five alternations of 400 steps on ONE box
cd "$(dirname "$0")/.." || exit 1
W=${1:-sparse-1m}
pj='import json,sys; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print(round(d["value"],1), round(d["roofline"]["avg_launch_ms"],4))'
for i in 1 2 3 4 5; do
a=$(timeout 600 python3 bench.py --no-cpu-baseline --steps 400 --workload $W 2>/dev/null | python3 -c "$pj")
b=$(DLG_TEST_LIB=$PWD/tools/micro/libvar.so timeout 600 python3 bench.py --no-cpu-baseline --steps 400 --workload $W 2>/dev/null | python3 -c "$pj")
echo "library: $a    libvar.so: $b"
done
