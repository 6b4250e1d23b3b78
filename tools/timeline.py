#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 kernel trace."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
idx = [i for i, r in enumerate(rows) if ('k_assemble(' in r['Kernel_Name'] or 'k_assemble_mfma' in r['Kernel_Name'])]
start = idx[-1]
seq = []
for r in rows[start:]:
    n = r['Kernel_Name']
    short = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    seq.append((short, dur, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    if short.startswith('k_norm2_Jv') and len(seq) > 20:
        break
for s, d, g, ts, te in seq:
    if s.startswith('k_factor'):
        print()
    print("%s[%d]=%.0f" % (s.replace('k_', '').replace('_level', '')[:11], g, d), end='  ')
print()
print("span %.0f us, sum of kernels %.0f us" % ((seq[-1][4] - seq[0][3]) / 1e3, sum(d for _, d, _, _, _ in seq)))
