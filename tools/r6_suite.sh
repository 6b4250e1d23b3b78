#!/bin/bash
# round 6: the whole GPU suite, then value with / without the pass over J on one box
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/${1:-r6s}; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $OUT/suite.txt
tail -8 $OUT/suite.txt
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench_a$i.json 2>$OUT/bench_a$i.err; python3 tools/pj.py < $OUT/bench_a$i.json
env DOGLEG_AMD_EI_JPASS=1 timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench_b$i.json 2>/dev/null; python3 tools/pj.py < $OUT/bench_b$i.json
done
