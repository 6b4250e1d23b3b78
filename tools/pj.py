import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["workload"], d["steps"], round(d["value"],1), d.get("separate_passes") and round(d["separate_passes"]["steps_per_s"],1), round(d["roofline"]["frac"],3), d["roofline"]["avg_launch_ms"])
