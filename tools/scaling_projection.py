#!/usr/bin/env python3
"""tools only: per-rank compute time of a partitioned step on ONE device (bench.py --logical-ranks R --rank r: the rank's
subtree partition, every sum over the ranks skipped) for r = 0 .. R-1, and the step time an R-GPU node would need under
stated all-reduce latencies.  A PROJECTION, labelled as such: nothing here ran on more than one GPU.
usage: tools/scaling_projection.py [--workload sparse-1m] [--ranks 8] [--steps 50] > profiles/rNN_scaling_projection.md"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sparse-1m")
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--lambda0", type=float, default=None, help="lambda of the one-GPU run too (config #5: 1.0, so that both sides factorise once)")
ap.add_argument("--rccl-floor", default=None, help="JSON of tools/rccl_floor.py: RCCL's measured cost per all-reduce at world size 1 on this box -- the first row of the table")
a = ap.parse_args()


def run(extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", a.workload, "--no-cpu-baseline", "--steps", str(a.steps),
                        "--warmup", "5"] + extra, capture_output=True, text=True, timeout=900)
    for l in r.stdout.splitlines():
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit("bench.py gave no line:\n" + r.stderr[-2000:])


one = run(["--lambda0", str(a.lambda0)] if a.lambda0 is not None else [])
rows = []
for r in range(a.ranks):
    d = run(["--logical-ranks", str(a.ranks), "--rank", str(r)])
    rows.append(d)
ph = ["K4_total", "K5_factor", "K6_solve", "K3K8_norm2Jv", "K7_step", "vec"]
print(f"# Scaling projection for `{a.workload}` over {a.ranks} ranks (one MI355X; NOT a measurement on {a.ranks} GPUs)\n")
print("Every rank's subtree partition run on one device with every sum over the ranks skipped (`bench.py --logical-ranks "
      f"{a.ranks} --rank r`, `dlg_backend_set_noop_comm`; lambda = 1 keeps the rank's partial top of the tree positive definite): the")
print("compute a rank does between the collectives, with the launch structure of the real partitioned step.  Milliseconds per step;")
print("phases from HIP events of a second loop (they overlap where streams overlap: K3 runs beside K5).\n")
print("| rank | rows | ms / step | " + " | ".join(ph) + " |")
print("|---|---|---|" + "---|" * len(ph))
for d in rows:
    lr = d["logical_rank"]
    p = d["phases_ms_per_step"]
    print(f"| {lr['rank']} | {lr['rows']} | {d['ms_per_step']:.4f} | " + " | ".join(f"{p.get(k, 0):.4f}" for k in ph) + " |")
p1 = one["phases_ms_per_step"]
print(f"| one GPU, no partition | {one['config']['Nmeas']} | {one['ms_per_step']:.4f} | " + " | ".join(f"{p1.get(k, 0):.4f}" for k in ph) + " |")
slow = max(d["ms_per_step"] for d in rows)
part = rows[0].get("partition") or {}
print(f"\nPartition: cut above level {part.get('cut_above_level')}, {part.get('replicated_supernodes')} replicated supernodes, "
      f"{part.get('bytes_summed_per_factorisation', 0)/1e6:.2f} MB summed per factorisation.\n")
N1 = one["config"]["Nstate"] + 1
big = max(8.0*N1, float(part.get("bytes_summed_per_factorisation", 0)))
print(f"Collectives per step (DESIGN §7): Jt*x + |x|^2 (N + 1 doubles = {8*N1/1e6:.2f} MB) at the evaluation; the cut buffer inside the factorisation; the")
print("solution + the Cauchy step's scalar (N + 1); |J step|^2 (1).  RCCL's all-reduce on 8 GPUs over xGMI was never measured here (one GPU per")
print("box) and its small-message latency is not in the guides: each all-reduce is ASSUMED to cost a latency plus 2 (R - 1) / R x bytes at 100 GB/s")
print("(a ring over point-to-point links of ~153 GB/s each).\n")
print("| assumed latency per all-reduce | projected ms / step | projected steps/s | against one GPU |")
print("|---|---|---|---|")
R = a.ranks
bw_ms = lambda nbytes: 2.0*(R - 1)/R*nbytes/100e9*1e3
wire = 2*bw_ms(8.0*N1) + bw_ms(float(part.get("bytes_summed_per_factorisation", 0))) + bw_ms(8.0)
floor = None
if a.rccl_floor and os.path.exists(a.rccl_floor):
    try:
        fl = json.loads([l for l in open(a.rccl_floor) if l.startswith("{")][0])      # (librccl prints its banner on stdout too)
        # two vectors of N + 1, the cut buffer, one scalar: what RCCL itself costs this stream with ONE rank (no wire, no peer)
        floor = (2*fl["N_plus_1_doubles_1.2MB"] + fl["cut_buffer_1.18MB"] + fl["1_double_8B"])*1e-3
        t = slow + floor
        print(f"| **measured floor**: RCCL at world size 1 on this box ({fl['N_plus_1_doubles_1.2MB']:.0f} / {fl['cut_buffer_1.18MB']:.0f} / {fl['1_double_8B']:.0f} us for 1.2 MB / the cut buffer / 8 bytes; no wire, no peer) | {t:.4f} | {1e3/t:.0f} | {one['ms_per_step']/t:.2f} x |")
    except Exception as ex:
        print(f"| (no usable {a.rccl_floor}: {ex}) | | | |")
for lat in (15e-3, 30e-3, 50e-3):
    t = slow + 4*lat + wire
    print(f"| {lat*1e3:.0f} us (assumed) + bytes at 100 GB/s | {t:.4f} | {1e3/t:.0f} | {one['ms_per_step']/t:.2f} x |")
k5_1, k5_r = p1.get("K5_factor", 0), max(d["phases_ms_per_step"].get("K5_factor", 0) for d in rows)
print(f"\nSlowest rank {slow:.4f} ms against {one['ms_per_step']:.4f} ms on one GPU: {one['ms_per_step']/slow:.2f} x is the ceiling the partition's compute "
      f"leaves before any collective is paid for ({wire*1e3:.0f} us of the projection are bytes on the wire).  The passes over J and the leaf level shrink with the")
print(f"rank's rows; the factorisation less so (K5 {k5_1:.3f} ms on one GPU, {k5_r:.3f} ms on the slowest rank): the levels between the leaves and the cut keep their")
print("per-level latency (one launch with replicas, `sparse_factor_setup`, region \"lo\"), then come pack / sum / unpack of the cut buffer and the replicated top")
print("of the tree.  Sharding rows does not shorten the elimination tree's critical path: a configuration whose step is mostly that path (config #4) gains")
print("nothing, one whose step is mostly passes over J and leaves (config #5) gains what those were.")
