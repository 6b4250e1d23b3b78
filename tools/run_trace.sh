#!/bin/bash
# tools only: kernel trace of a short bench run -> one step's critical-queue timeline (tools/step_gaps.py)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-trace}; rm -rf "$O"; mkdir -p "$O"
shift
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" > $O/bench_tr.json 2>$O/err.txt
python3 tools/step_gaps.py $O/tr 20 > $O/gaps.txt 2>&1
find $O/tr -name '*.csv' -size +1M -delete
cat $O/gaps.txt
