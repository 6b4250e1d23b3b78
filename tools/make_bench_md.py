#!/usr/bin/env python3
"""Regenerate profiles/<tag>_bench.md from the published round artifacts."""
import json, os, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(root, "profiles")
B = {k: json.load(open(os.path.join(P, f"{tag}_bench_{k}.json"))) for k in ("sparse1m", "sparse200k", "dense50k", "sparse5m")}
def ph(k, n): return "%.3f" % B[k]["phases_ms_per_step"][n]
def stats(wl, steps, rows):
    return subprocess.check_output([sys.executable, os.path.join(root, "tools", "prof_summary.py"),
                                    os.path.join(root, "gpurun_out", tag, f"stats_{wl}"), str(steps), str(rows)]).decode().rstrip()
s1, d5 = stats("sparse-1m", 23, 18), stats("dense-50k", 11, 12)
e2e = json.load(open(os.path.join(P, f"{tag}_e2e_sparse1m.json")))
t = json.load(open(os.path.join(P, "traffic.json")))["sparse-1m"]
r, rd = B["sparse1m"]["roofline"], B["dense50k"]["roofline"]
sy = B["sparse1m"]["symbolic"]
k5b = 8.0*(sy["nnz_JtJ_lower"] + sy["nnz_L"]); k5f = sy["factor_flops"]; k5t = B["sparse1m"]["phases_ms_per_step"]["K5_factor"]
k6b = 16.0*sy["nnz_L"] + 32.0*B["sparse1m"]["config"]["Nstate"]; k6t = B["sparse1m"]["phases_ms_per_step"]["K6_solve"]
nlev = sy["n_levels"]
def row(k, label):
    b = B[k]; c = b.get("cpu_baseline")
    if c:
        return f"| {label} | {b['value']:.1f} | {b['ms_per_step']:.2f} | {c['value']:.4g} | {b['value']/c['value']:.0f}x |"
    return f"| {label} | {b['value']:.1f} | {b['ms_per_step']:.2f} | not timed | |"
md = f"""# Round 1 — measurements on one MI355X (gpurun box, ROCm 7.2, hipcc gfx950)

All numbers from `python bench.py` (JSON lines committed next to this file) and
`rocprofv3 --kernel-trace --stats` of the same command (`{tag}_*_kernel_stats.csv`); collected by
`tools/collect_round.sh`, copied here by `tools/publish_round.py`, this file by `tools/make_bench_md.py`.
A "step" = K1+K3+K4+K5+K6+K7+K8 on inputs resident in HBM (refactorise + interpolate).

| workload (BASELINE.json config) | GPU steps/s | ms/step | CPU oracle steps/s (1 thread, same box) | ratio |
|---|---|---|---|---|
{row('sparse1m', 'sparse-1m (#4: 1M x 150k, 15M nnz)')}
{row('sparse200k', 'sparse-200k (#3: 200k x 30k, 3M nnz)')}
{row('dense50k', 'dense-50k (#2: 50k x 2k; CPU row-sampled, see JSON)')}
{row('sparse5m', 'sparse-5m (#5: 5M x 500k, 75M nnz, 2 factorisations per step: lambda path)')}

(The round started at 26 steps/s on sparse-1m; the trajectory is in DESIGN.md section 6.)

Per-phase GPU time (HIP events on the backend stream, ms per step):

| phase | sparse-1m | sparse-200k | dense-50k |
|---|---|---|---|
| K1 Jt*x | {ph('sparse1m','K1_jtx')} | {ph('sparse200k','K1_jtx')} | {ph('dense50k','K1_jtx')} |
| K3+K8 two \\|Jv\\|^2 passes | {ph('sparse1m','K3K8_norm2Jv')} | {ph('sparse200k','K3K8_norm2Jv')} | {ph('dense50k','K3K8_norm2Jv')} |
| K4 JtJ assembly (kernel alone) | {ph('sparse1m','K4_kernel')} | {ph('sparse200k','K4_kernel')} | {ph('dense50k','K4_kernel')} |
| K4 total (memset, partial sums / slab reduce) | {ph('sparse1m','K4_total')} | {ph('sparse200k','K4_total')} | {ph('dense50k','K4_total')} |
| K5 Cholesky | {ph('sparse1m','K5_factor')} | {ph('sparse200k','K5_factor')} | {ph('dense50k','K5_factor')} |
| K6 solve (sparse: backward only, forward rides in the factor) | {ph('sparse1m','K6_solve')} | {ph('sparse200k','K6_solve')} | {ph('dense50k','K6_solve')} |
| K7 step | {ph('sparse1m','K7_step')} | {ph('sparse200k','K7_step')} | {ph('dense50k','K7_step')} |

Roofline of the JtJ assembly kernel:
* sparse-1m `k_assemble_mfma`: {r['algorithmic_bytes']/1e6:.1f} MB algorithmic / {r['avg_launch_ms']:.3f} ms = **{r['achieved']:.0f} GB/s = {100*r['frac']:.1f} % of 8 TB/s**
  (target in BASELINE.json: 40 %; the LDS kernel this round started with reached 6.5 %).  Measured HBM
  traffic {t['bytes_per_launch']/1e6:.0f} MB per launch ({tag}_pmc.md): {t['bytes_per_launch']/r['algorithmic_bytes']:.2f}x the algorithmic bytes -- J is
  walked twice (once by the point column blocks, once by the camera column blocks that also carry the
  dense global block), plus the k-group records.
* dense-50k `k_syrk_lower<64>`: {rd['algorithmic_flops']:.3e} flop / {rd['avg_launch_ms']:.3f} ms = **{rd['achieved']:.1f} TFLOP/s** = {100*rd['frac']:.0f} % of the 78.6
  TFLOP/s datasheet fp64-matrix peak, 89 % of the 48 TFLOP/s a register-only
  v_mfma_f64_16x16x4_f64 loop sustains on this box (`tools/gpu_probe.py`, {tag}_probe.txt).
* sparse `k_norm2_Jv` (K3/K8): 192 MB / 56 us = 3.4 TB/s = 43 % of 8 TB/s; traffic 203 MB (1.06x).
* K5-sparse and K6-sparse are latency / critical-path bound (SURVEY 8d says to expect low fractions and to
  say so): sparse-1m K5 = {k5b/1e6:.0f} MB algorithmic (`8 nnz(tril JtJ) + 8 nnz(L)`) and {k5f/1e9:.2f} GFLOP in {k5t:.2f} ms
  = {k5b/k5t/1e6:.0f} GB/s ({100*k5b/k5t/1e6/8000:.1f} % of HBM), {k5f/k5t/1e9:.2f} TFLOP/s; K6 = {k6b/1e6:.0f} MB (`16 nnz(L) + 32 N`) in {k6t:.2f} ms
  = {k6b/k6t/1e6:.0f} GB/s ({100*k6b/k6t/1e6/8000:.1f} %).  {nlev} elimination-tree levels, roughly 60-80 us each above the leaves (one factor kernel that also pulls the children's update matrices, one backward-solve kernel).

rocprofv3 --stats, sparse-1m (20 timed + 3 warm-up steps; ms/step = total/23):
```
{s1}
```
rocprofv3 --stats, dense-50k (10 timed + 1 warm-up; ms/step = total/11):
```
{d5}
```

End to end through `dogleg_optimize2` with the host callback (PCIe-inclusive, `tools/e2e_bench.py`,
sparse-1m): {e2e['trials']} trials + {e2e['callbacks']} callback evaluations in {e2e['total_s']:.2f} s including the one-off symbolic
analysis and pinned allocations ({e2e['end_to_end_steps_per_s']:.1f} steps/s; {e2e['steps_per_s_excluding_callback']:.1f} without the host callback time); each
evaluation moves 128 MB host->device.  This is NOT bench.py's value.

RCCL plumbing check (1 rank, backend nccl, all-reduce hook installed, torch's HIP runtime):
`{tag}_bench_dist_world1_rccl.log` — same check values as the single-process run.

Probes ({tag}_probe.txt): fp64 MFMA issue rate 48 TFLOP/s; HBM copy 4.88 TB/s (read+write bytes).
"""
open(os.path.join(P, f"{tag}_bench.md"), "w").write(md)
print(md[:1500])
