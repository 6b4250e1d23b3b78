#!/usr/bin/env python3
"""bench.py -- trust-region steps/sec (fp64) of the dog-leg hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A "step" is one full trial step of the hot path on an operating point whose x
and Jacobian values are already resident in HBM: K1 (Jt*x, |x|^2) -> K3 (Cauchy)
-> K4+K5 (JtJ assembly + Cholesky) -> K6 (Gauss-Newton solve) -> K7 (dog-leg
interpolation, p_new back to the host) -> K8 (expected improvement), driven
through the C-ABI of libdogleg_amd.so exactly as the host trust-region driver
drives it (including its host<->device scalar round trips).  This is the
expensive kind of step (refactorisation + interpolation); cached retries after
a rejection are cheaper.

Workloads (BASELINE.json configs):
  sparse-1m   (default) config #4: BA-arrowhead 1 000 000 meas x 150 000 params, 15 M nnz
  sparse-200k config #3: 200 000 x 30 000, 3 M nnz
  dense-50k   config #2: dense 50 000 x 2 000
  sparse-5m   config #5: 5 000 000 x 500 001, 75 M nnz, ill-conditioned + lambda path

N > 1 (launched by torch.distributed.run, one process per GPU): sparse workloads use the SUBTREE
PARTITION of the elimination tree (include/dlg_backend.h: every rank holds the measurement rows of
its subtrees, assembles / factors / solves them alone; per step the ranks sum Jt*x, one small buffer
at the cut of the tree -- 1.2 MB on config #4 over 8 ranks --, the solution and two scalars); the
dense workload shards contiguous rows and sums JtJ.  The sums are RCCL all-reduces enqueued by the
library on its own stream (dlg_backend_init_rccl; torch.distributed only hands out the unique id,
the barrier and the max over ranks of the elapsed time).  The problem size is fixed: "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (kind, params)
    "sparse-1m":   ("sparse", dict(Nc=2499, Np=45000, Nobs=500000)),
    "sparse-200k": ("sparse", dict(Nc=499, Np=9000, Nobs=100000)),
    "sparse-5m":   ("sparse", dict(Nc=8333, Np=149999, Nobs=2500000, scale_decades=4.0, n_zero_cols=3)),
    "dense-50k":   ("dense", dict(M=50000, N=2000)),
    "sparse-tiny": ("sparse", dict(Nc=49, Np=900, Nobs=10000)),
    "dense-tiny":  ("dense", dict(M=3000, N=256)),
}

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F64_PEAK_TFLOPS = 78.6    # public datasheet; a register-only v_mfma_f64_16x16x4_f64 loop sustains 72 - 78 here (profiles/r04_probe.txt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="sparse-1m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--copies", type=int, default=3, help="resident copies of (x, J) the timed loop rotates over")
    # a PROJECTION input, not a bench line: rank --rank of a partition over --logical-ranks ranks on this one device, every
    # sum over the ranks skipped (dlg_backend_set_noop_comm) -- the rank's compute time per phase
    ap.add_argument("--logical-ranks", type=int, default=1)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--lambda0", type=float, default=None, help="lambda of every step (logical ranks: 1.0, the partial top of the tree must stay positive definite)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    use_dist = world > 1 or os.environ.get("DLG_BENCH_FORCE_DIST") == "1"
    logical = args.logical_ranks > 1 and not use_dist
    prank, pworld = (args.rank, args.logical_ranks) if logical else (rank, world)
    lam0 = args.lambda0 if args.lambda0 is not None else (1.0 if logical else 0.0)

    torch = dist = None
    if use_dist:
        # torch.distributed (backend nccl = RCCL) before any GPU call of this process; libdogleg_amd
        # then binds to the HIP runtime torch ships
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import numpy as np
    from libdogleg_amd import capi
    import problems as oa                    # the synthetic problem generators (problems/); the oracle is loaded by the cpu_baseline leg only

    L = capi.lib()
    if L.dlg_device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")

    kind, prm = WORKLOADS[args.workload]
    t_setup = time.time()
    rccl_ranks = None
    if kind == "sparse":
        prob = oa.BAProblem(**prm, seed=11)
        N, M, nnz = prob.N, prob.M, prob.nnz
        Jp, Ji = prob.pattern()
        p0 = prob.p0()
        x, Jx = prob.eval(p0)
        be = capi.Backend(capi.DLG_SPARSE, N, M, nnz, device=local_rank if use_dist else -1)
    else:
        prob = oa.DenseProblem(**prm, seed=11)
        N, M, nnz = prob.N, prob.M, 0
        p0 = prob.p0()
        x, J = prob.eval(p0)
        be = capi.Backend(capi.DLG_DENSE, N, M, device=local_rank if use_dist else -1)
    if use_dist:
        # the library's own RCCL communicator: rank 0 creates the id, torch.distributed hands it out
        idt = torch.zeros(128, dtype=torch.uint8, device=torch.device("cuda", local_rank))
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.rccl_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        be.init_rccl(rank, world, bytes(idt.cpu().numpy().tobytes()))
        rccl_ranks = be.comm_size()
        assert rccl_ranks == world, f"RCCL reports {rccl_ranks} ranks, expected {world}"
    part = None
    if logical:
        be.set_noop_comm(True)
    if kind == "sparse":
        if use_dist or logical:
            be.set_partition(prank, pworld)
        be.set_pattern(Jp, Ji)
        sym = be.stats()
        if use_dist or logical:
            rows = be.partition_rows()
            part = be.partition_stats()
            x_loc = x[rows]
            J_loc = np.concatenate([Jx[Jp[r]:Jp[r+1]] for r in rows]) if len(rows) else np.zeros(0)
            row0, row1 = 0, len(rows)
        else:
            row0, row1 = 0, M
            x_loc, J_loc = x, Jx
    else:
        sym = {}
        # contiguous row ranges
        row0 = (M * prank) // pworld
        row1 = (M * (prank + 1)) // pworld
        x_loc, J_loc = x[row0:row1], J[row0:row1]
        if use_dist or logical:
            be.set_shard(row0, row1, None)
    # The timed loop rotates over NCOPY resident copies of (x, J): one unchanging J of 120 MB (config #4)
    # would sit in the 256 MiB Infinity Cache between steps and flatter every "HBM" fraction
    # (MI355X_MICROARCH.md: scale past L3 before reading bandwidth numbers); 3 copies = 360 MB+ do not.
    ncopy = max(1, args.copies)
    xh, Jh = np.ascontiguousarray(x_loc), np.ascontiguousarray(J_loc)
    d_x = [capi.DeviceArray(xh) for _ in range(ncopy)]
    d_J = [capi.DeviceArray(Jh) for _ in range(ncopy)]
    be.set_p(0, p0)
    setup_s = time.time() - t_setup

    state = {"tr": None, "i": 0}

    def one_step():
        c = state["i"] % ncopy
        state["i"] += 1
        be.bind_device(0, d_x[c].ptr, d_J[c].ptr)    # a fresh operating point: nothing cached
        norm2x, gmax = be.eval(0)                    # K1
        # K3..K8 as the driver issues them for a fresh point once steps leave the trust region's edge
        # behind (driver.hip take_step -> dlg_take_step): Cauchy step, compute_updateGN (factorise from
        # lambda = 0 with the reference's lambda loop, dogleg.c:656-677, and solve), the choice of the kind
        # of step, the step, its expected improvement, p_new D2H -- one synchronisation per attempt.
        # The trust region is known before the step, as in the driver (state["tr"], from the first step).
        if state["tr"] is None:
            lam, n2c, n2g = be.cauchy_gauss_newton(0, lam0)
            tr = 0.5 * (n2c ** 0.5 + n2g ** 0.5)
            n2s, k, amax, ei, pnew = be.step(0, 1, capi.KIND_INTERP, tr)
            state["tr"] = tr
        else:
            lam, r, pnew = be.take_step(0, 1, state["tr"], lam0)
            assert logical or r["kind"] == capi.KIND_INTERP
            n2c, n2g, n2s, k, amax, ei = r["n2c"], r["n2g"], r["n2s"], r["k"], r["amax"], r["ei"]
        return norm2x, n2c, n2g, k, n2s, ei, gmax, amax, lam

    def run_steps(n):
        """n steps as ONE call into the library (dlg_run_steps: the same bind / eval / take_step sequence, the
        host side in C as in the library's own driver): the timed loop does not carry the interpreter's
        overhead per entry point (~25 us of a 0.73 ms step on config #4)"""
        if state["tr"] is None:
            one_step()
            n -= 1
        if n <= 0:
            return None
        r, kind = be.run_steps(0, 1, n, [d.ptr for d in d_x], [d.ptr for d in d_J], state["i"] % ncopy, state["tr"], lam0)
        state["i"] += n
        assert logical or kind == capi.KIND_INTERP
        return r["n2x"], r["n2c"], r["n2g"], r["k"], r["n2s"], r["ei"], r["gmax"], r["amax"], r["lam"]

    def barrier():
        L.dlg_device_sync()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()

    # Evaluation of a point = Jt*x AND JtJ in ONE pass over J (dlg_backend_set_speculation(1), as the driver
    # runs once steps need the Gauss-Newton step: sparse_eval_assemble, the assembly kernel's B operand
    # times x); the factorisation of the step adopts those panels.  All of K1..K8 is inside the timed
    # loop either way; the same loop with the two passes apart is reported as "separate_passes".
    one_pass = kind == "sparse"
    if one_pass:
        be.set_speculation(True)
        # ... and the expected improvement's pass over J (K8) behind the decision point (dlg_backend_set_defer_tail, as the
        # driver runs device-callback solves): the reference uses the value behind the NEXT evaluation (dogleg.c:1427), so
        # K8 of step i runs beside the evaluation of step i + 1 and dlg_run_steps fetches it there (dlg_step_tail);
        # the last step's tail is waited for inside the timed region.  `inline_tail` below: K8 in line, as in rounds 1-4
        if not use_dist and not logical:
            be.set_defer_tail(True)
    elif kind == "dense" and not use_dist and not logical:
        be.set_defer_tail(True)         # (the same for the dense pass: K8 and p_new behind the step kernel the host waits for)
    res = one_step()
    for _ in range(args.warmup):
        res = one_step()
    # Untimed steps until the device has worked for a quarter of a second (at least 30 steps): the first steps behind
    # the allocations run 5 - 10 % slower whatever W is, and the first process on a fresh box was once seen at 0.72 ms a
    # step for a whole timed region with every phase at its usual time.  Not part of the K steps.
    # (N > 1: every rank must run the SAME number of steps -- the steps hold collectives --, so the count is fixed there)
    presteps, t_pre = 0, time.perf_counter()
    n_pre = 30 if kind == "sparse" else 10
    while presteps < 30 or (not use_dist and time.perf_counter() - t_pre < 0.25) or (use_dist and presteps < 6 * n_pre):
        run_steps(n_pre)
        presteps += n_pre
        if presteps >= 3000:
            break
    # in the timed loop only the roofline kernel is bracketed by events (two records per step); the table
    # of all phases comes from a second loop: twenty event records per step cost ~6 % of a 1.1 ms step
    # (... and only every fourth launch of it: a kernel whose completion somebody listens to holds the next dispatch
    # back by ~5 us -- an artefact of measuring that sat inside `value` in every step; the average below is over the
    # sampled launches of the timed region)
    be.set_profiling(True, only=["K4_kernel"], every=4 if args.steps >= 8 else 1)
    barrier()
    t0 = time.perf_counter()
    res = run_steps(args.steps) or res
    barrier()
    elapsed = time.perf_counter() - t0
    prof_k4 = be.profile()
    be.set_profiling(True)
    run_steps(args.steps)
    barrier()
    prof = be.profile()
    prof["K4_kernel"] = prof_k4["K4_kernel"]
    # launches that returned after their first barrier behind a failed factorisation (the lambda path of config #5:
    # what is left of K5, K6 and K8 of the attempt at lambda = 0) are NOT in `prof`: per-launch averages below are
    # over launches that ran in full
    prof_early = be.profile_early()
    be.set_profiling(False)
    # the cheaper kind of trial step, reported beside `value`: after a rejected step the
    # reference re-uses the cached Cauchy/GN steps and the factor (dogleg.c:533-535, 637, 825), so a
    # retry is only step formation (K7), expected improvement (K8) and the evaluation of the new
    # point (K1).  Not part of `value`.
    retry_ms = first_retry_ms = None
    if not use_dist and not logical:
        tr_retry = 0.5 * (res[1] ** 0.5 + res[2] ** 0.5)

        def one_retry(shrink):
            # (the expected improvement is fetched where the driver needs it: behind the evaluation of the new trial point)
            be.step(0, 1, capi.KIND_INTERP, tr_retry * shrink, tail=False)
            c = state["i"] % ncopy
            state["i"] += 1
            be.bind_device(1, d_x[c].ptr, d_J[c].ptr)
            be.eval(1)
            be.step_tail()
        one_retry(0.99)
        barrier()
        tr0 = time.perf_counter()
        for i in range(args.steps):
            one_retry(0.98 - 1e-4 * i)
        barrier()
        retry_ms = (time.perf_counter() - tr0) / args.steps * 1e3
        # ... and the FIRST retry behind an accepted step: the evaluation of the trial point had enqueued the leaf level of
        # its factorisation ahead (step_prepare), the rejection abandons it (sparse_abandon_enqueued); the retries of a
        # run of rejections (above) find nothing enqueued
        t_first = 0.0
        n_first = max(8, args.steps // 4)
        for i in range(n_first + 2):
            one_step()
            c = state["i"] % ncopy
            state["i"] += 1
            be.bind_device(1, d_x[c].ptr, d_J[c].ptr)
            be.eval(1)
            ta = time.perf_counter()
            one_retry(0.98 - 1e-4 * i)
            if i >= 2:
                t_first += time.perf_counter() - ta
        first_retry_ms = t_first / n_first * 1e3
        one_step()
    # A workload whose factorisation breaks down at lambda = 0 (config #5: exactly-zero columns): `value` restarts every step
    # from lambda0 = 0 -- the worst case (rounds 1-4: K4 + the failed K5 of the first attempt paid every time; round 5: a look at
    # the diagonal in front of the attempt, sparse_host.hip).  The reference's lambda is
    # sticky (dogleg.c:138, 670-673): after the first failure a real solve starts every later step at the lambda that worked.
    sticky = None
    if kind == "sparse" and not use_dist and not logical and res[8] != lam0:
        lam_s = res[8]
        rs0 = be.run_steps(0, 1, max(3, args.warmup), [d.ptr for d in d_x], [d.ptr for d in d_J], state["i"] % ncopy, state["tr"], lam_s)
        barrier()
        tq0 = time.perf_counter()
        rs, kd = be.run_steps(0, 1, args.steps, [d.ptr for d in d_x], [d.ptr for d in d_J], state["i"] % ncopy, state["tr"], lam_s)
        barrier()
        sticky_ms = (time.perf_counter() - tq0) / args.steps * 1e3
        assert rs["lam"] == lam_s and abs(rs["n2s"] - res[4]) <= 1e-9 * abs(res[4])
        sticky = {"ms_per_step": sticky_ms, "steps_per_s": 1e3 / sticky_ms, "lambda": lam_s,
                  "what": "the same step started at the lambda the first step ended with (the reference's sticky lambda): one factorisation a step; "
                          "`value` starts every step at lambda0: since round 5 that costs a look at the diagonal and one more host "
                          "synchronisation (the doomed attempt enqueues nothing, its panels are factored at the next lambda: no second assembly)"}
    # the same step with K1 and K4 as two passes over J (how rounds 1-2 reported `value`)
    sep_ms = None
    if one_pass and not logical:
        be.set_speculation(False)
        one_step()
        barrier()
        ts0 = time.perf_counter()
        res_s = run_steps(args.steps)
        barrier()
        sep_ms = (time.perf_counter() - ts0) / args.steps * 1e3
        be.set_speculation(True)
        assert abs(res_s[4] - res[4]) <= 1e-9 * abs(res[4]), "the one-pass evaluation changed the step"
    # the same step with the expected improvement IN FRONT of the synchronisation the host decides behind (what a host-callback
    # caller -- the literal drop-in -- gets: it needs p_new on the host before it can evaluate the trial point, so nothing of
    # the step can hide behind its way back): `value` is the pipelined device-callback form (VERDICT r5 "weak" 10)
    inline_ms = None
    ei_src = None
    if not use_dist and not logical and kind in ("sparse", "dense"):
        ei_src = be.ei_source()
        be.set_defer_tail(False)
        one_step()
        barrier()
        ti0 = time.perf_counter()
        res_i = run_steps(args.steps)
        barrier()
        inline_ms = (time.perf_counter() - ti0) / args.steps * 1e3
        be.set_defer_tail(True)
        assert abs(res_i[4] - res[4]) <= 1e-9 * abs(res[4]) and abs(res_i[5] - res[5]) <= 1e-9 * abs(res[5]), "the in-line form changed the step"
    if use_dist:
        dev = torch.device("cuda", local_rank)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the JtJ assembly kernel (the kernel north_star sets a target for)
    ms, cnt = prof["K4_kernel"]
    k4_ms = ms / max(cnt, 1)
    if kind == "sparse":
        # 12*nnz + 4*(M+1) + 8*nnz(tril JtJ) bytes per launch (local rows on a shard)
        nnz_loc = int(J_loc.shape[0])
        alg_bytes = 12 * nnz_loc + 4 * (row1 - row0 + 1) + 8 * sym["nnz_JtJ_lower"]
        if one_pass:
            alg_bytes += 8 * (row1 - row0) + 8 * N          # + x in, Jt*x out: K1 rides in the same pass
        roof = {"kernel": ("k_assemble_mfma<.,true> (K1+K4-sparse: Jt*x and JtJ in one pass over J)" if one_pass
                           else "k_assemble_mfma (K4-sparse JtJ assembly)"), "bound": "hbm",
                "achieved": alg_bytes / (k4_ms * 1e-3) / 1e9 if k4_ms > 0 else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                "algorithmic_bytes": alg_bytes, "avg_launch_ms": k4_ms, "launches": cnt}
    else:
        flops = 2.0 * (row1 - row0) * N * (N + 1) / 2
        roof = {"kernel": "k_syrk_lower (K4-dense fp64 MFMA SYRK)", "bound": "mfma",
                "achieved": flops / (k4_ms * 1e-3) / 1e12 if k4_ms > 0 else None,
                "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None,
                "algorithmic_flops": flops, "avg_launch_ms": k4_ms, "launches": cnt}
    roof["frac"] = (roof["achieved"] / roof["peak"]) if roof["achieved"] else None
    # HBM traffic of that kernel from a separate rocprofv3 --pmc pass (FETCH_SIZE, WRITE_SIZE in
    # their own runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 -- the
    # factor was re-checked on this repo's 8 B/lane streaming kernels, see profiles/r01_pmc.md).
    # bench.py cannot collect PMC counters itself; the committed numbers are per launch.
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        ent = tj.get(args.workload) if world == 1 else None
        if ent:
            roof["traffic"] = ent["bytes_per_launch"]
            roof["traffic_source"] = ent["source"]
    except Exception:
        pass

    # the other phases against their own bounds (SURVEY 8d formulas): the factorisation and the
    # triangular solves are latency / critical-path bound -- low fractions are expected, they are
    # printed so that they can be watched
    others = {}
    def per_launch(name):
        ms_, cnt_ = prof[name]
        return ms_ / max(cnt_, 1)
    if kind == "sparse":
        k5_ms, k6_ms = per_launch("K5_factor"), per_launch("K6_solve")
        k5_bytes = 8 * sym["nnz_JtJ_lower"] + 8 * sym["nnz_L"]
        k6_bytes = 16 * sym["nnz_L"] + 32 * N
        jv_ms = per_launch("K3K8_norm2Jv")
        jv_bytes = 12 * int(J_loc.shape[0]) + 4 * (row1 - row0) + 8 * N
        others = {
            "K5_sparse_cholesky": {"bound": "latency (critical path of the elimination tree)", "ms": k5_ms,
                                   "algorithmic_bytes": k5_bytes, "GBps": k5_bytes / (k5_ms * 1e-3) / 1e9 if k5_ms > 0 else None,
                                   "frac_hbm": k5_bytes / (k5_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k5_ms > 0 else None,
                                   "flops": sym["factor_flops"], "TFLOPps": sym["factor_flops"] / (k5_ms * 1e-3) / 1e12 if k5_ms > 0 else None,
                                   "frac_mfma": sym["factor_flops"] / (k5_ms * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS if k5_ms > 0 else None,
                                   "levels": sym["n_levels"]},
            "K6_sparse_solve": {"bound": "latency", "ms": k6_ms, "algorithmic_bytes": k6_bytes,
                                "GBps": k6_bytes / (k6_ms * 1e-3) / 1e9 if k6_ms > 0 else None,
                                "frac_hbm": k6_bytes / (k6_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k6_ms > 0 else None},
            "K3K8_norm2_Jv": {"bound": "hbm", "ms": jv_ms, "algorithmic_bytes": jv_bytes,
                              "GBps": jv_bytes / (jv_ms * 1e-3) / 1e9 if jv_ms > 0 else None,
                              "frac_hbm": jv_bytes / (jv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if jv_ms > 0 else None},
        }
    else:
        k5_ms, k6_ms = per_launch("K5_factor"), per_launch("K6_solve")
        others = {
            "K5_dense_potrf": {"bound": "latency (one launch; critical path = the diagonal tiles' panel sweeps)", "ms": k5_ms, "flops": N**3 / 3.0,
                               "TFLOPps": N**3 / 3.0 / (k5_ms * 1e-3) / 1e12 if k5_ms > 0 else None,
                               "frac_mfma": N**3 / 3.0 / (k5_ms * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS if k5_ms > 0 else None},
            "K6_dense_potrs": {"bound": "latency (one launch; hand-off per 64 rows and sweep)", "ms": k6_ms, "algorithmic_bytes": 2 * 8 * N * (N + 1) // 2,
                               "GBps": 8.0 * N * (N + 1) / (k6_ms * 1e-3) / 1e9 if k6_ms > 0 else None},
        }

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "trust-region steps/sec (fp64) at fixed (Nmeas,Nstate,nnz)",
            "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "kind": kind, "Nmeas": M, "Nstate": N, "nnz": nnz,
                       "step": "K1+K3+K4+K5+K6+K7+K8 (refactorise + interpolate), inputs resident in HBM"
                               + ("; K1 and K4 share one pass over J" if one_pass else ""),
                       "parallelism": (("subtree partition" if kind == "sparse" else "row sharding") + f" x{world}, RCCL in-stream") if use_dist else
                                      (f"rank {prank} of {pworld} LOGICAL ranks on one device, every sum over the ranks skipped: per-rank compute time for a projection, not a throughput" if logical else "1 GPU")},
            "roofline": roof,
            "other_kernels": others,
            "untimed_presteps": presteps,
            "inputs": {"resident_copies": ncopy, "bytes_per_copy": int(xh.nbytes + Jh.nbytes),
                       "note": "the timed loop rotates over the copies: past the 256 MiB Infinity Cache"},
            "phases_ms_per_step": {k: (v[0] + prof_early[k][0]) / args.steps for k, v in prof.items()},
            "early_return_launches": ({k: {"launches": v[1], "ms_per_launch": v[0] / v[1]} for k, v in prof_early.items() if v[1] > 0}
                                      or None),
            "cached_retry_step": ({"ms_per_step": retry_ms, "steps_per_s": 1e3 / retry_ms,
                                   "first_retry_after_an_accepted_step_ms": first_retry_ms,
                                   "what": "K7 + K8 + evaluation of the new point (K1, with K4 in the same pass where `value` has it), cached Cauchy/GN/factor; "
                                           "ms_per_step: inside a run of rejections (nothing enqueued ahead); first_retry...: the retry right behind an accepted "
                                           "step, whose trial point's leaf-level factorisation had been enqueued ahead and is abandoned"}
                                  if retry_ms else None),
            "sticky_lambda_step": sticky,
            "expected_improvement": ({"placement": "behind the decision point",
                                      "from_solved_system": (ei_src[0] if ei_src else None), "pivot_ratio": (ei_src[1] if ei_src else None),
                                      "what": "dlg_backend_set_defer_tail (as the library's device-callback solves run): every step's expected improvement and its "
                                              "p_new copy are INSIDE the timed region, enqueued behind the step kernel the host waits for; the value is fetched "
                                              "behind the next evaluation, where rho needs it (dogleg.c:1410-1427; the last step's inside the region too).  "
                                              "from_solved_system: |J step|^2 came from (JtJ) gn = -Jt x instead of a pass over J (lambda = 0, pivot ratio "
                                              "<= 212: include/dlg_backend.h; DOGLEG_AMD_EI_JPASS=1: always the pass).  `inline_tail`: the value in front of the "
                                              "synchronisation, as in rounds 1-4"}
                                     if (not use_dist and not logical and kind in ("sparse", "dense"))
                                     else {"placement": "in front of the step's synchronisation"}),
            "inline_tail": ({"ms_per_step": inline_ms, "steps_per_s": 1e3 / inline_ms,
                             "what": "the same step with the expected improvement and p_new in front of the synchronisation the host decides "
                                     "behind (dlg_backend_set_defer_tail off): what a host-callback caller of dogleg_optimize2 gets from the "
                                     "hot path, its own callback and the H2D of J aside"} if inline_ms else None),
            "separate_passes": ({"ms_per_step": sep_ms, "steps_per_s": 1e3 / sep_ms,
                                 "what": "same step with Jt*x (K1) and the JtJ assembly (K4) as two passes over J: `value` of rounds 1-2"}
                                if sep_ms else None),
            "symbolic": sym, "setup_s": setup_s, "rccl_ranks": rccl_ranks,
            "partition": ({"cut_above_level": part["cut_level"], "replicated_supernodes": part["supernodes_above_cut"],
                           "rows_rank0": part["rows_mine"], "bytes_summed_per_factorisation": 8 * part["reduced_doubles"],
                           "bytes_panel_buffer": 8 * part["panel_doubles"]} if part else None),
            "check": {"norm2_x": res[0], "norm2_step": res[4], "expected_improvement": res[5], "lambda": res[8]},
        }

    # ---- CPU baseline: the oracle's restatement of the same step, host cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not logical:
        from tests import oracle_api         # (the CPU oracle: this leg and nothing else in bench.py)
        O = oracle_api.oracle()
        from libdogleg_amd.ctypes_defs import dptr, iptr
        work = np.zeros(5 * N)
        o8 = np.zeros(8)
        if kind == "sparse":
            ta = time.perf_counter()
            F = O.orc_sparse_analyze(N, M, iptr(Jp), iptr(Ji))
            t_an = time.perf_counter() - ta
            n_done, t_cpu = 0, 0.0
            while n_done < 1 or (t_cpu < args.cpu_seconds and n_done < args.steps):
                tb = time.perf_counter()
                rc = O.orc_step_sparse(F, N, M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(x), dptr(p0),
                                       res[8], dptr(work), dptr(o8))
                t_cpu += time.perf_counter() - tb
                n_done += 1
                assert rc == 0
            sample = (f"{n_done} full steps of the same workload, single thread; sparse Cholesky = up-looking simplicial, "
                      f"static minimum-degree ordering (a CHOLMOD-simplicial stand-in; symbolic analysis {t_an:.1f}s excluded; "
                      f"nnz(L)={O.orc_sparse_nnzL(F)} against {sym['nnz_L']} of the GPU's nested dissection, "
                      f"factor flops={O.orc_sparse_flops(F):.3g})")
            O.orc_sparse_free(F)
            cpu_val = n_done / t_cpu
            ocheck = o8.copy()
        else:
            # the faithful rank-1 JtJ loop costs ~M*N^2/2 FMAs: time it on a row sample and scale
            Ms = min(M, max(64, int(args.cpu_seconds * 1.2e9 / (N * N / 2 + 3 * N))))
            dfac = np.zeros(N * (N + 1) // 2)
            tb = time.perf_counter()
            rc = O.orc_step_dense(N, Ms, dptr(J[:Ms]), dptr(x[:Ms]), dptr(p0), 1e-3, dptr(dfac),
                                  dptr(work), dptr(o8))
            t_s = time.perf_counter() - tb
            # split: Cholesky+solve cost does not scale with M; estimate it separately
            tb = time.perf_counter()
            O.orc_dpptrf_L(N, dptr(dfac))
            t_chol = time.perf_counter() - tb
            t_full = (t_s - t_chol) * (M / Ms) + t_chol
            cpu_val = 1.0 / t_full
            sample = (f"1 step on the first {Ms} of {M} rows ({t_s:.1f}s), row-proportional part "
                      f"scaled to {M} rows; packed Cholesky ({t_chol:.2f}s) not scaled; single thread")
            ocheck = None
        out["cpu_baseline"] = {"value": cpu_val, "unit": "steps/s", "cores": 1, "kind": "port",
                               "sample": sample, "host_cpus": os.cpu_count()}
        out["speedup_vs_cpu_baseline"] = out["value"] / cpu_val
        if ocheck is not None:
            # same inputs -> same numbers: a cheap end-to-end parity check printed with the result
            out["check"]["oracle_norm2_step"] = float(ocheck[4])
            out["check"]["oracle_expected_improvement"] = float(ocheck[5])
            out["check"]["rel_diff_norm2_step"] = abs(float(ocheck[4]) - res[4]) / max(1e-300, abs(res[4]))
    if rank == 0:
        if logical:
            out["metric"] = "per-rank compute time of one partitioned step (projection input; NOT trust-region steps/sec)"
            out["value"] = None
            out["logical_rank"] = {"rank": prank, "ranks": pworld, "lambda0": lam0, "rows": int(row1 - row0)}
        print(json.dumps(out))
    be.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
