#!/usr/bin/env python3
"""Copy the summaries tools/collect_round.sh produced (gpurun_out/<tag>/) into profiles/ and
rebuild profiles/traffic.json from the PMC passes.  usage: tools/publish_round.py r01"""
import json, os, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
names = {"sparse-1m": "sparse1m", "sparse-200k": "sparse200k", "dense-50k": "dense50k", "sparse-5m": "sparse5m"}
for wl, short in names.items():
    f = os.path.join(src, f"bench_{wl}.json")
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_bench_{short}.json"))
    st = os.path.join(src, f"stats_{wl}", "p_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(dst, f"{tag}_{short}_kernel_stats.csv"))
for a, b in (("e2e_sparse1m.json", f"{tag}_e2e_sparse1m.json"), ("bench_dist_world1_rccl.log", f"{tag}_bench_dist_world1_rccl.log"),
             ("probe.txt", f"{tag}_probe.txt"), ("sq_k4.txt", f"{tag}_sq_k_assemble_mfma.txt"), ("k4_split.txt", f"{tag}_k4_split.txt"),
             ("top_of_tree_levels.txt", f"{tag}_top_of_tree_levels.txt"), ("e2e.err", f"{tag}_e2e_timing.txt"),
             ("scaling_projection.md", f"{tag}_scaling_projection.md"), ("rccl_floor.json", f"{tag}_rccl_floor.json"), ("trsv_hops.txt", f"{tag}_trsv_hops.txt"), ("potrf_diag.txt", f"{tag}_potrf_diag.txt"), ("panel_sweep.txt", f"{tag}_panel_sweep.txt"), ("xcd_probe.txt", f"{tag}_xcd_probe.txt"), ("scaling_projection_sparse5m.md", f"{tag}_scaling_projection_sparse5m.md"), ("step_trace.txt", f"{tag}_step_trace.txt"),
             ("sq_syrk.txt", f"{tag}_sq_k_syrk_lower.txt"), ("top_of_tree_levels_config5.txt", f"{tag}_top_of_tree_levels_config5.txt")):
    if os.path.exists(os.path.join(src, a)):
        # (profiles/r04_sq_k_assemble_mfma.txt was published EMPTY and a text computed from it -- an empty source
        # is a failed collection step, not an artifact)
        if os.path.getsize(os.path.join(src, a)) == 0:
            sys.exit(f"publish_round: {os.path.join(src, a)} is empty -- the step of tools/collect_round.sh that writes it failed; "
                     "re-collect (or delete the file if the round does not carry it)")
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))


def pmc(wl, counter, kernel):
    f = os.path.join(src, f"pmc_{wl}_{counter}.txt")
    best = None
    for line in open(f):
        if kernel in line.split(" dispatches")[0]:
            m = re.search(r"'%s': ([0-9.]+)" % counter, line)
            if m:
                best = float(m.group(1))
    return best


traffic = {}
# the dense JtJ launch is one of many k_syrk_lower<64> dispatches (the potrf trailing updates use the
# same kernel): its entry is maintained by hand from the per-dispatch CSV (see the pmc notes)
for wl, kernel, label in (("sparse-1m", "k_assemble_mfma<18, true, true>", "k_assemble_mfma<18, true, true> (K1+K4 in one pass)"),):
    try:
        f, w = pmc(wl, "FETCH_SIZE", kernel), pmc(wl, "WRITE_SIZE", kernel)
        if f is None or w is None:
            continue
        traffic[wl] = {"kernel": label, "fetch_size_kb": f, "write_size_kb": w, "bytes_per_launch": int((2*f + w)*1024),
                       "source": f"profiles/{tag}_pmc.md: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; "
                                 "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts half of streamed reads)"}
    except FileNotFoundError:
        pass
# dense: since the factorisation is one launch of its own (k_potrf_tiles), every k_syrk_lower<64> dispatch is a JtJ launch
try:
    f, w = pmc("dense-50k", "FETCH_SIZE", "k_syrk_lower<64>"), pmc("dense-50k", "WRITE_SIZE", "k_syrk_lower<64>")
    fr, wr = pmc("dense-50k", "FETCH_SIZE", "k_syrk_reduce<128>"), pmc("dense-50k", "WRITE_SIZE", "k_syrk_reduce<128>")
    if f is not None and w is not None:
        traffic["dense-50k"] = {"kernel": "k_syrk_lower<64> (JtJ launch, split-K slabs)", "fetch_size_kb": f, "write_size_kb": w,
                                "bytes_per_launch": int((2*f + w)*1024),
                                "k_syrk_reduce_bytes_per_launch": int((2*(fr or 0) + (wr or 0))*1024),
                                "source": f"profiles/{tag}_pmc.md: as above; every k_syrk_lower<64> dispatch of the run is a JtJ launch "
                                          "(the factorisation is k_potrf_tiles); FETCH_SIZE counts L2-to-fabric requests, Infinity-Cache hits included"}
except FileNotFoundError:
    pass
old = {}
try:
    old = json.load(open(os.path.join(dst, "traffic.json")))
except Exception:
    pass
old.update(traffic)
json.dump(old, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
for wl in ("sparse-1m", "dense-50k"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(src, f"pmc_{wl}_{c}.txt")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f"{tag}_pmc_{names[wl]}_{c}.txt"))
print(json.dumps(traffic, indent=1))
