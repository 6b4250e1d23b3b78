#!/usr/bin/env python3
"""Idle time of the GPU inside a bench step, from a rocprofv3 kernel trace: per step (delimited by
the K1 kernel k_jtx / dense equivalent) the wall span, the kernel-busy time and the largest gaps."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if ('k_jtx(' in r['Kernel_Name'] or 'k_jtx_rows(' in r['Kernel_Name'])]
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:30]
segs = [(a, b) for a, b in zip(starts, starts[1:]) if int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']) > 1000000]
for a, b in segs[-2:]:
    seg = rows[a:b]
    t0, t1 = int(seg[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    gaps = []
    for x, y in zip(seg, seg[1:] + [rows[b]]):
        g = int(y['Start_Timestamp']) - int(x['End_Timestamp'])
        gaps.append((g, short(x['Kernel_Name']), short(y['Kernel_Name'])))
    gaps.sort(reverse=True)
    print("step: wall %.0f us, busy %.0f us, idle %.0f us" % ((t1 - t0)/1e3, busy/1e3, (t1 - t0 - busy)/1e3))
    for g, x, y in gaps[:8]:
        print("   gap %6.1f us  after %-30s before %s" % (g/1e3, x, y))
