#!/bin/bash
# tools only: kernel statistics of config #5 (sparse-5m)
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s5m; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py --workload sparse-5m --steps 10 --warmup 2 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/s5m/stats/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print("%-60s calls %6s avg %9.1f us  %5.1f %%" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
find $O -name "*kernel_trace.csv" -delete
