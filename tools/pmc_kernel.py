#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch for kernels matching a name."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if pat and pat not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(cnt[k])
    print(k, "dispatches", n, {c: round(v / n, 1) for c, v in acc[k].items()})
