#!/bin/bash
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> <bench args...>; separate passes for each counter
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out.$c.log 2>&1
  python tools/pmc_kernel.py $out/$c | grep -E "assemble|jtx|norm2_Jv|update|factor|solve"
done
