#!/bin/bash
# tools only: tools/ab.sh KNOB OUTDIR [notests] -- the GPU suite, then `value` with / without one knob on ONE box, then a step trace
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
K=$1; OUT=gpurun_out/${2:-ab}; mkdir -p $OUT
if [ "$3" != notests ]; then
timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -40 > $OUT/suite.txt
tail -12 $OUT/suite.txt
fi
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench_a$i.json 2>$OUT/bench_a$i.err; python3 tools/pj.py < $OUT/bench_a$i.json
env $K=1 timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench_b$i.json 2>/dev/null; python3 tools/pj.py < $OUT/bench_b$i.json
done
timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-200k > $OUT/bench_c.json 2>/dev/null; python3 tools/pj.py < $OUT/bench_c.json
env $K=1 timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-200k > $OUT/bench_d.json 2>/dev/null; python3 tools/pj.py < $OUT/bench_d.json
rm -rf $OUT/tr; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 > $OUT/bench_tr.json 2>$OUT/err_tr.txt
python3 tools/step_gaps.py $OUT/tr 20 > $OUT/step_trace.txt 2>&1
find $OUT/tr -name '*.csv' -size +1M -delete
cat $OUT/step_trace.txt
python3 - <<PY
import json
for n in ("a1","b1"):
    d=json.loads([l for l in open("$OUT/bench_%s.json" % n) if l.startswith("{")][0])
    print(n, round(d["value"],1), "inline", d.get("inline_tail") and round(d["inline_tail"]["steps_per_s"],1), "retry", d["cached_retry_step"] and round(d["cached_retry_step"]["ms_per_step"],4), d["expected_improvement"].get("from_solved_system"), d["check"])
PY
