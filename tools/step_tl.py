#!/usr/bin/env python3
"""tools only: the kernel timeline of the k-th evaluation-to-evaluation step of a rocprofv3 kernel trace (csv dir, k)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
k = int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ts = [i for i, r in enumerate(rows) if 'k_assemble_mfma<18, true, true>' in r['Kernel_Name']]
d = [(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']))/1000 for a, b in zip(ts, ts[1:])]
print("step lengths:", [round(x) for x in d])
i0, i1 = ts[k], ts[k+1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1+1]:
    print("%8.1f %8.1f  q%s  %s" % ((int(r['Start_Timestamp']) - t0)/1000, (int(r['End_Timestamp']) - t0)/1000, r['Queue_Id'], r['Kernel_Name'][:60]))
