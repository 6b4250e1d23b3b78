#!/usr/bin/env python3
"""profiles/<tag>_pmc.md and <tag>_e2e.md from the round's collected files (tools/collect_round.sh, tools/publish_round.py):
the tables are generated, not kept by hand.  usage: tools/make_round_md.py r04 [previous-tag]"""
import json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
prev = sys.argv[2] if len(sys.argv) > 2 else "r%02d" % (int(tag[1:]) - 1)
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles") + "/"


def rd(f, c):
    d = {}
    for l in open(P + f):
        m = re.match(r"(?:void )?(.+?) dispatches (\d+) \{'%s': ([0-9.]+)\}" % c, l.strip())
        if m:
            d[m.group(1)] = float(m.group(3))
    return d


F, W = rd(f"{tag}_pmc_sparse1m_FETCH_SIZE.txt", "FETCH_SIZE"), rd(f"{tag}_pmc_sparse1m_WRITE_SIZE.txt", "WRITE_SIZE")
b = json.load(open(P + f"{tag}_bench_sparse1m.json"))
bd = json.load(open(P + f"{tag}_bench_dense50k.json"))
alg = b["roofline"]["algorithmic_bytes"]
rows = [("**`k_assemble_mfma<18, true, true>` (K1+K4 in one pass: the roofline kernel of `value`)**", "k_assemble_mfma<18, true, true>", alg),
        ("`k_assemble_mfma<18, false, false>` (K4 alone, the `separate_passes` loop)", "k_assemble_mfma<18, false, false>", 265.8e6),
        ("`k_norm2_Jv` (K3/K8)", "k_norm2_Jv", 185.2e6),
        ("`k_factor_level<256, true>` (leaf level)", "k_factor_level<256, true>", None),
        ("`k_update_gather<256>`", "k_update_gather<256>", None),
        ("`k_factor_level<512, false>` (the upper levels, one launch)", "k_factor_level<512, false>", None),
        ("`k_solve_bwd_level<256, true>` (leaves)", "k_solve_bwd_level<256, true>", None),
        ("`k_clear_ranges` (the panels above the merged leaves)", "k_clear_ranges", None),
        ("`k_touch` (the leaf panels primed for the backward solve)", "k_touch", None)]
out = [f"# Round {int(tag[1:])} — PMC counters (rocprofv3 --pmc, separate passes with --kernel-trace only)\n",
       "## Traffic (FETCH_SIZE / WRITE_SIZE)\n",
       f"`tools/collect_round.sh {tag}`: `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py",
       "--steps 3 --warmup 1 --no-cpu-baseline` and the same with `--pmc WRITE_SIZE`; per-dispatch averages by",
       f"`tools/pmc_kernel.py` (`{tag}_pmc_*_{{FETCH,WRITE}}_SIZE.txt`).  HBM bytes = (2·FETCH_SIZE + WRITE_SIZE)·1024",
       "(gfx950: FETCH_SIZE reports one half of streamed reads, MI355X_MICROARCH.md; `r01_pmc.md`).  The bench loop",
       "rotates over three resident copies of (x, J): past the 256 MiB Infinity Cache.  This file: `tools/make_round_md.py`.\n",
       "| kernel (sparse-1m, per launch) | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes | algorithmic bytes | ratio |", "|---|---|---|---|---|---|"]
for label, k, a in rows:
    f, w = F.get(k), W.get(k)
    if f is None:
        continue
    hb = (2*f + w)*1024
    out.append(f"| {label} | {f:,.0f} | {w:,.0f} | {hb/1e6:.1f} MB | {('%.1f MB' % (a/1e6)) if a else '—'} | {('%.2f' % (hb/a)) if a else ''} |".replace(",", " "))
t = json.load(open(P + "traffic.json"))["dense-50k"]
rf = bd["roofline"]
out.append("\nThe assembly kernel moves 1.6 × its algorithmic bytes (J walked once by the point columns' tasks and again by the camera columns'):")
out.append("unchanged since round 2.  What bounds it is not that traffic: without any load of J and without its transient stores the kernel still")
out.append(f"takes 83 of its ≈ 104 µs (`{tag}_experiments.md`, ablations).\n")
out.append(f"Dense (dense-50k, per launch): `k_syrk_lower<64>` (the JtJ launch) FETCH {t['fetch_size_kb']:,.0f} KB, WRITE {t['write_size_kb']:,.0f} KB = {t['bytes_per_launch']/1e9:.2f} GB".replace(",", " "))
out.append("across the fabric for 0.82 GB of algorithmic bytes (Infinity-Cache hits are counted by these counters) — 11.9 GB in round 3 and at the")
out.append("start of round 4: the split-major, XCD-aware order of the workgroups lets the tiles of one row range share J's rows in L2.  The kernel")
out.append(f"runs at {rf['achieved']:.1f} TFLOP/s algorithmic = {100*rf['frac']:.0f} % of the matrix cores' fp64 peak (the true peak: `{tag}_probe.txt`, `{tag}_probe_notes.md`).")
out.append(f"`k_syrk_reduce<128>` {t['k_syrk_reduce_bytes_per_launch']/1e6:.1f} MB.  `traffic.json` carries both entries from this round.\n")
out.append("## SQ counters of the assembly kernel\n")
out.append(f"`tools/run_sq.sh` (three passes of eight counters over `tools/k4_split.py`), `{tag}_sq_k_assemble_mfma.txt`:\n\n```")
out += [l.rstrip() for l in open(P + f"{tag}_sq_k_assemble_mfma.txt")]
out.append("```")
open(P + f"{tag}_pmc.md", "w").write("\n".join(out) + "\n")

e = json.load(open(P + f"{tag}_e2e_sparse1m.json"))
hc, dc = e["host_callback"], e["device_callback"]
# the library's own breakdown (DOGLEG_AMD_TIMING=1): one block of "timing:" lines per solve; the device-callback solves
# are the ones whose callback is an enqueue (microseconds), the untraced ones record nothing per trial
blocks, cur = [], []
for l in open(P + f"{tag}_e2e_timing.txt"):
    if "timing:" not in l:
        continue
    cur.append(l.rstrip().split("timing:", 1)[1])
    if "teardown" in l:
        blocks.append(cur); cur = []


def val(block, key):
    for l in block:
        if key in l:
            m = re.search(r"([0-9.]+) ms", l)
            return float(m.group(1)) if m else None
    return None


dev_plain = [b_ for b_ in blocks if (val(b_, "model callback") or 1e9) < 1.0 and (val(b_, "trace / vnlog") or 0.0) == 0.0 and val(b_, "dlg_point_eval") is not None]
bk = dev_plain[-1] if dev_plain else []
neval = dc.get("evaluations_per_solve", e["callbacks"])
plain_s = dc.get("untraced_solve_s_median", dc["second_call_s"])
plain_rate = dc.get("untraced_steps_per_s", dc["steps_per_s"])
md = f"""# Round {int(tag[1:])} — end to end through the public API (NOT `bench.py`'s `value`)

`DOGLEG_AMD_TIMING=1 python tools/e2e_bench.py --workload sparse-1m` on one MI355X (config #4: 1M × 150k,
15M non-zeros; {e['trials']} trial steps, {neval} evaluations **per solve**).  JSON: `{tag}_e2e_sparse1m.json`; the library's own
breakdown on stderr: `{tag}_e2e_timing.txt`.  This file: `tools/make_round_md.py`.

| solve | wall | trial steps/s |
|---|---|---|
| host callback, first solve of the process (symbolic analysis, code objects, allocations) | {hc['first_call_s']:.3f} s | |
| host callback, next solve | **{hc['second_call_s']:.4f} s** | {hc['steps_per_s']:.1f} ({e['callbacks']} × {hc['callback_s_each']*1e3:.0f} ms of callback + {e['callbacks']} × {hc['h2d_bytes_per_eval']/1e6:.0f} MB over PCIe) |
| device callback (`dogleg_optimize_device2`), as a user calls it (median of five) | **{plain_s:.4f} s** | **{plain_rate:.0f}** |
| device callback with the tests' per-trial trace (every record downloads the step vector behind its own synchronisation) | {dc['second_call_s']:.4f} s | {dc['steps_per_s']:.0f} |
| device callback, 20 iterations allowed (traced) | {e['device_callback_20_iterations']['solve_s']:.4f} s | {e['device_callback_20_iterations']['steps_per_s']:.0f} |

Where one device-callback solve spends its time (host clock around the driver's own calls, the last untraced solve of the run):

```
{chr(10).join(l.strip() for l in bk)}
```

max |p_device − p_host| = {e['max_abs_p_diff_device_vs_host']:.1e}.  The end-to-end rate stays out of `value`: `bench.py` times the hot path on inputs
resident in HBM ({b['value']:.0f} steps/s on this workload in the same collection).  A `dlg_point_eval` here waits for the model's own kernel
too (the callback only enqueues it on the backend's stream): `problems/device_problems.hip`'s block-arrowhead model took 0.65 ms an
evaluation in round 4 — one thread a row, a row's entries 120 bytes apart — and was two thirds of the "unaccounted" time of a solve
(VERDICT r4 #6); read with lane = entry and summed per row out of LDS in the host's order (same bits) it takes ≈ 0.1 ms.  With
a host callback the solver gets {hc['steps_per_s']:.0f} trial steps/s (the callback and PCIe: what a re-linked libdogleg user gets).
"""
open(P + f"{tag}_e2e.md", "w").write(md)
print("wrote", f"{tag}_pmc.md", f"{tag}_e2e.md")
