#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a short per-kernel table."""
import csv, glob, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
print("%-46s %6s %12s %10s %10s" % ("kernel", "calls", "ms/step", "avg_us", "max_us"))
for r in list(csv.DictReader(open(f)))[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-46s %6s %12.3f %10.1f %10.1f" % (n[:46], r["Calls"], float(r["TotalDurationNs"]) / steps / 1e6,
                                               float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
