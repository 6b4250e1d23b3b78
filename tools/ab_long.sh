#!/bin/bash
# tools only: tools/ab_long.sh KNOB [workload] -- `value` with / without one knob, five alternations of 400 steps on ONE box
cd "$(dirname "$0")/.." || exit 1
K=$1; W=${2:-sparse-1m}
for i in 1 2 3 4 5; do
a=$(timeout 600 python3 bench.py --no-cpu-baseline --steps 400 --workload $W 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))")
b=$(env $K=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 400 --workload $W 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))")
echo "default: $a    $K=1: $b"
done
