import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); ph=d.get("phases_ms_per_step") or {}
        print(d["config"]["workload"], d["steps"], round(d["value"],1), d.get("separate_passes") and round(d["separate_passes"]["steps_per_s"],1), round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_ms"],4),
              "K5", round(ph.get("K5_factor",0),4), "K6", round(ph.get("K6_solve",0),4))
