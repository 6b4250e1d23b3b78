#!/usr/bin/env python3
"""tools only: per-level summary of the persistent factor launch from the lines tools/prof_factor.sh prints
(DLG_FL_DUMP_ALL=1): when the children of a level's workgroups had been added, when their update matrices were done,
and the phases of the workgroup that finished last."""
import re, sys, statistics as st
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s+wg\s+(\d+) \(w\s+(\d+) rows\s+(\d+) nch (\d+) u_lds (\d)\): start\s+(\d+) children there\s+(\d+) "
                 r"panel in\s+(\d+) added\s+(\d+) factored\s+(\d+) tail\s+(\d+) flag\+stored\s+(\d+)", l)
    if m:
        rows.append(tuple(int(x) for x in m.groups()))
# (the dump may hold several launches: keep the last record of every workgroup)
last = {}
for r in rows:
    last[r[0]] = r
rows = sorted(last.values())
# levels by arrival time of the children: a new level starts where 'children there' jumps
lv, cur, last = [], [], None
for r in rows:
    if last is not None and r[6] > last + 2500 and len(cur) > 0 and r[6] > max(x[6] for x in cur) - 100:
        pass
    cur.append(r); last = r[6]
n = len(rows); sizes = []; k = n
while k > 0:
    s = (k + 1)//2; sizes.append(s); k -= s
b = [0]
for s in sizes: b.append(b[-1] + s)
for k in range(len(b) - 1):
    g = [r for r in rows if b[k] <= r[0] - rows[0][0] < b[k+1]]
    if not g: continue
    late = max(g, key=lambda r: r[10])
    print(f"level +{k}: {len(g):3d} wg  children added min/med/max {min(r[6] for r in g)}/{int(st.median([r[6] for r in g]))}/{max(r[6] for r in g)}"
          f"  W done min/med/max {min(r[10] for r in g)}/{int(st.median([r[10] for r in g]))}/{max(r[10] for r in g)}"
          f"  last: wg {late[0]} w {late[1]} rows {late[2]} add {late[8]-late[6]} factor {late[9]-late[8]} tail {late[10]-late[9]} store {late[11]-late[10]}  (10 ns)")
