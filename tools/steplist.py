#!/usr/bin/env python3
"""Kernel list of one bench step (k_jtx to the next k_jtx) from a rocprofv3 kernel trace."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if ('k_jtx(' in r['Kernel_Name'] or 'k_jtx_rows(' in r['Kernel_Name'])]
segs = [(a, b) for a, b in zip(starts, starts[1:]) if int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']) > 1000000]
a, b = segs[-2]
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:24]
prev = None; busy = 0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3
    g = (int(r['Start_Timestamp']) - prev)/1e3 if prev else 0
    prev = int(r['End_Timestamp']); busy += d
    print("%-26s %7.1f  gap %5.1f  grid %s" % (short(r['Kernel_Name']), d, g, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])))
print("wall %.0f us, busy %.0f us, %d launches" % ((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']))/1e3, busy, b - a))
