#!/usr/bin/env python3
"""critical-stream timeline of one bench step from a rocprofv3 kernel trace (main stream only)"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:34]
# steps delimited by the assembly kernel
idx = [i for i, r in enumerate(rows) if 'k_assemble_mfma' in r['Kernel_Name']]
if not idx:        # a dense workload: a step begins with the pass that forms Jt*x
    idx = [i for i, r in enumerate(rows) if 'k_gemvT_part' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -3
a, b = idx[k], idx[k+1]
seg = rows[a:b]
# keep the stream (queue) of the assembly kernel
q = seg[0].get('Queue_Id') or seg[0].get('Stream_Id')
key = 'Queue_Id' if 'Queue_Id' in seg[0] else 'Stream_Id'
t0 = int(seg[0]['Start_Timestamp'])
prev_end = t0
tot_gap = 0
for r in seg + [rows[b]]:
    if r[key] != q: 
        print("      (other queue) %-34s %7.1f .. %7.1f" % (short(r['Kernel_Name']), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3)); continue
    g = (int(r['Start_Timestamp']) - prev_end)/1e3
    tot_gap += max(g, 0)
    print("gap %6.1f  %-34s %7.1f .. %7.1f  (%.1f)" % (g, short(r['Kernel_Name']), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
    prev_end = int(r['End_Timestamp'])
print("total gaps on the critical queue: %.1f us" % tot_gap)
