#!/usr/bin/env python3
"""tools only: the kernels of one steady-state step of bench.py (sparse) from a rocprofv3 --kernel-trace csv,
in start order, with start / end relative to the step's assembly kernel and the queue they ran on."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_assemble_mfma' in r['Kernel_Name'] and 'true' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
i0, i1 = idx[k], idx[k+1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0-3:i1-3]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:90]}")
print("step period:", (int(rows[i1]['Start_Timestamp']) - t0)/1000)
