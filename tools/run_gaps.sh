#!/bin/bash
# tools only: kernel trace of a short bench run -> one step's critical-queue timeline (tools/step_gaps.py), then bench lines
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/gaps; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_on -o t -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/bench_tr.json 2>$O/err_on.txt
python3 tools/step_gaps.py $O/tr_on 20 > $O/gaps_on.txt 2>&1
find $O/tr_on -name '*.csv' -size +1M -delete
cat $O/gaps_on.txt
timeout 900 python3 -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "ahead or take_step or speculative or fused" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2; do
timeout 300 python3 bench.py --no-cpu-baseline > $O/bench_on_$i.json 2>>$O/err.txt; python3 tools/pj.py < $O/bench_on_$i.json
done
