#!/usr/bin/env python3
"""tools only: per-level summary of the persistent factor launch from the lines tools/prof_factor.sh prints
(DLG_FL_DUMP_ALL=1): when the children of a level's workgroups had been added, when the panel sweep was over, when
the update matrices were done and handed over, for replica 0 of the supernode that finished last and the spread
over the level (10 ns units)."""
import re, sys, statistics as st
rows = {}
for l in open(sys.argv[1]):
    m = re.match(r"\s+wg\s+(\d+) \(w\s+(\d+) rows\s+(\d+) nch (\d+) u_lds (\d) lvl (\d+) rep (\d+)\): start\s+(\d+) flag up\s+(\d+) "
                 r"panel in\s+(\d+) added\s+(\d+) factored\s+(\d+) tail\s+(\d+) flag\+stored\s+(\d+)", l)
    if m:
        v = tuple(int(x) for x in m.groups())
        # (the dump may hold several launches: keep the last record of every workgroup -- of the one-launch region, levels >= 1,
        # and of the leaf-level launch, level 0, apart: until round 5 they shared the key, and what was printed as "level 0:
        # 321 supernodes" were the leaf launch's workgroups 703 .. 1023 that no region workgroup had overwritten)
        rows[(v[5] == 0, v[0])] = v
lv = {}
for v in rows.values():
    lv.setdefault(v[5], []).append(v)
prev = None
for l in sorted(lv):
    g = lv[l]
    if l == 0:
        print(f"(level 0 is the leaf level's own launch, k_factor_level<256, true>: {len(g)} of its workgroups recorded; the one-launch region is levels 1 and up)")
        continue
    nrep = max(v[6] for v in g) + 1
    late = max(g, key=lambda v: v[8])
    wdone = [v[8] for v in g]            # flag up
    line = (f"level {l}: {len(g)//nrep:3d} supernodes x {nrep} replicas  children added med/max {int(st.median([v[10] for v in g]))}/{max(v[10] for v in g)}"
            f"  W out min/med/max {min(wdone)}/{int(st.median(wdone))}/{max(wdone)}"
            f"  last: wg {late[0]} w {late[1]} rows {late[2]} rep {late[6]}: wait+add {late[10]-late[9]} sweep {late[11]-late[10]} tail {late[12]-late[11]} hand-off {late[8]-late[12]} panel store {late[13]-late[8]}")
    if prev is not None:
        line += f"  | level time {max(wdone) - prev}"
    prev = max(wdone)
    print(line)
