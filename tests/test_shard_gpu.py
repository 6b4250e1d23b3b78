"""Row sharding on the GPU with k logical ranks on ONE device (gpurun boxes have a single
GPU): every rank is a backend of its own with dlg_backend_set_shard, driven by its own host thread;
the all-reduce hook is an in-process sum (download, barrier, add in rank order, upload) standing in
for RCCL.  Every rank must end up with the oracle's step."""
import ctypes as C
import threading
import numpy as np
import pytest

from libdogleg_amd import capi
from libdogleg_amd.ctypes_defs import dptr, iptr
from tests import oracle_api as oa

pytestmark = pytest.mark.gpu


class _InProcessAllReduce:
    def __init__(self, world):
        self.world, self.slots = world, [None] * world
        self.bar = threading.Barrier(world)
        self.L = capi.lib()

    def hook(self, rank):
        def fn(buf, count, cookie):
            try:
                host = np.empty(count)
                if self.L.dlg_mem_download(host.ctypes.data, buf, 8 * count) != 0:
                    return 1
                self.slots[rank] = host
                self.bar.wait(timeout=120)
                total = self.slots[0].copy()
                for r in range(1, self.world):          # fixed order: every rank gets the same bits
                    total += self.slots[r]
                self.bar.wait(timeout=120)
                return 0 if self.L.dlg_mem_upload(buf, total.ctypes.data, 8 * count) == 0 else 1
            except Exception as e:                       # never let an exception cross the C boundary
                print("in-process all-reduce failed:", e)
                return 1
        return fn


def _sharded_step(kind, N, M, nnz, pattern, x, Jvals, p, cuts, row_slice, one_pass=False):
    world = len(cuts) - 1
    ar = _InProcessAllReduce(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            r0, r1 = cuts[rank], cuts[rank + 1]
            be = capi.Backend(kind, N, M, nnz)
            be.set_shard(r0, r1, ar.hook(rank))
            if pattern is not None:
                be.set_pattern(*pattern)
                be.set_speculation(one_pass)
            be.set_p(0, p)
            be.upload(0, x[r0:r1], row_slice(r0, r1))
            n2x, gmax = be.eval(0)
            n2c = be.cauchy(0)
            assert be.factorize(0, 0.0)
            n2g = be.solve_gn(0)
            tr = 0.5 * (np.sqrt(n2c) + np.sqrt(n2g))
            n2s, k, amax, pnew = be.make_step(0, 1, capi.KIND_INTERP, tr)
            ei = be.expected_improvement(0, 1)
            out[rank] = dict(n2x=n2x, n2c=n2c, n2g=n2g, k=k, ei=ei, step=be.download(1, capi.VEC_STEP),
                             gn=be.download(0, capi.VEC_GN))
            be.close()
        except Exception as e:
            errs.append((rank, repr(e)))
            try:
                ar.bar.abort()
            except Exception:
                pass

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs, errs
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_sparse_rows_sharded_over_logical_ranks(gpu, world):
    O = oa.oracle()
    prob = oa.BAProblem(49, 900, 10000, seed=7)
    N, M, nnz = prob.N, prob.M, prob.nnz
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    # uneven cuts, one of them inside an observation's pair of rows
    cuts = [0] + [int(M * (i + 1) / world) + (1 if i == 0 else 0) for i in range(world - 1)] + [M]
    res = _sharded_step(capi.DLG_SPARSE, N, M, nnz, (Jp, Ji), x, Jx, p, cuts, lambda a, b: Jx[Jp[a]:Jp[b]])
    # the oracle's step on the full problem
    F = O.orc_sparse_analyze(N, M, iptr(Jp), iptr(Ji))
    work, o8 = np.zeros(5 * N), np.zeros(8)
    assert O.orc_step_sparse(F, N, M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(x), dptr(p), 0.0, dptr(work), dptr(o8)) == 0
    O.orc_sparse_free(F)
    step_ref = work[3 * N:4 * N]
    for r in res:
        assert np.linalg.norm(r["step"] - step_ref) <= 1e-10          # tolerance of the parity bar
        assert abs(r["n2x"] - o8[0]) <= 1e-12 * o8[0]
        assert abs(r["ei"] - o8[5]) <= 1e-9 * abs(o8[5])
    for r in res[1:]:                                                  # replicated factor: identical bits
        assert np.array_equal(r["step"], res[0]["step"]) and r["k"] == res[0]["k"]


def test_dense_rows_sharded_over_two_logical_ranks(gpu):
    prob = oa.DenseProblem(3000, 256, seed=5)
    M, N = prob.M, prob.N
    p = prob.p0()
    x, J = prob.eval(p)
    J = np.ascontiguousarray(J).reshape(-1)
    cuts = [0, 1301, M]
    res = _sharded_step(capi.DLG_DENSE, N, M, 0, None, x, J, p, cuts, lambda a, b: J[a * N:b * N])
    full = _sharded_step(capi.DLG_DENSE, N, M, 0, None, x, J, p, [0, M], lambda a, b: J[a * N:b * N])[0]
    for r in res:
        assert np.linalg.norm(r["step"] - full["step"]) <= 1e-10
        assert abs(r["n2x"] - full["n2x"]) <= 1e-12 * full["n2x"]
    assert np.array_equal(res[0]["step"], res[1]["step"])


# ------------------------------------------------------------ subtree partition -------
def _partition_step(prob, world, use_take_step=False, lam0=0.0, one_pass=False):
    """every logical rank: backend with dlg_backend_set_partition + the in-process sum as the
    all-reduce hook; x / J of the rows the symbolic phase gave it; one full trial step"""
    N, M, nnz = prob.N, prob.M, prob.nnz
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    ar = _InProcessAllReduce(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            be = capi.Backend(capi.DLG_SPARSE, N, M, nnz)
            be.set_partition(rank, world)
            be.set_allreduce(ar.hook(rank))
            be.set_pattern(Jp, Ji)
            be.set_speculation(one_pass)       # JtJ assembled in the pass over the rank's rows that forms its share of Jt*x
            rows = be.partition_rows()
            st = be.partition_stats()
            Jloc = np.concatenate([Jx[Jp[r]:Jp[r+1]] for r in rows]) if len(rows) else np.zeros(0)
            be.set_p(0, p)
            be.upload(0, np.ascontiguousarray(x[rows]), Jloc)
            n2x, gmax = be.eval(0)
            if use_take_step:
                # the one-synchronisation op of the driver, sharded: first a plain step to learn the trust region
                lam, n2c, n2g = be.cauchy_gauss_newton(0, lam0)
                tr = 0.5 * (np.sqrt(n2c) + np.sqrt(n2g))
                be.upload(0, np.ascontiguousarray(x[rows]), Jloc)
                be.eval(0)
                lam, r, pnew = be.take_step(0, 1, tr, lam0)
                n2c, n2g, k, ei = r["n2c"], r["n2g"], r["k"], r["ei"]
            else:
                n2c = be.cauchy(0)
                lam, n2g = be.gauss_newton(0, lam0)
                tr = 0.5 * (np.sqrt(n2c) + np.sqrt(n2g))
                n2s, k, amax, pnew = be.make_step(0, 1, capi.KIND_INTERP, tr)
                ei = be.expected_improvement(0, 1)
            out[rank] = dict(n2x=n2x, n2c=n2c, n2g=n2g, k=k, ei=ei, lam=lam, step=be.download(1, capi.VEC_STEP),
                             gn=be.download(0, capi.VEC_GN), g=be.download(0, capi.VEC_JTX), rows=rows, stats=st)
            be.close()
        except Exception as e:
            errs.append((rank, repr(e)))
            try:
                ar.bar.abort()
            except Exception:
                pass

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not errs, errs
    return out, (N, M, Jp, Ji, x, Jx, p)


def _check_partition_against_oracle(res, data, lam=0.0, tol=1e-10):
    O = oa.oracle()
    N, M, Jp, Ji, x, Jx, p = data
    F = O.orc_sparse_analyze(N, M, iptr(Jp), iptr(Ji))
    work, o8 = np.zeros(5 * N), np.zeros(8)
    assert O.orc_step_sparse(F, N, M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(x), dptr(p), lam, dptr(work), dptr(o8)) == 0
    O.orc_sparse_free(F)
    step_ref, gn_ref = work[3 * N:4 * N], work[2 * N:3 * N]
    # the rows of the ranks are a partition of the measurements
    allrows = np.sort(np.concatenate([r["rows"] for r in res]))
    assert np.array_equal(allrows, np.arange(M))
    worst = 0.0
    for r in res:
        d = np.linalg.norm(r["step"] - step_ref)
        worst = max(worst, d)
        assert d <= tol, d
        assert np.linalg.norm(r["gn"] - gn_ref) <= tol
        assert abs(r["n2x"] - o8[0]) <= 1e-12 * o8[0]
        assert abs(r["n2c"] - o8[1]) <= 1e-10 * o8[1]
        assert abs(r["ei"] - o8[5]) <= 1e-9 * abs(o8[5])
    for r in res[1:]:          # the top of the tree is replicated and every sum is the same on every rank: identical bits
        assert np.array_equal(r["step"], res[0]["step"]) and np.array_equal(r["gn"], res[0]["gn"]) and r["k"] == res[0]["k"]
        assert np.array_equal(r["g"], res[0]["g"])
    return worst


@pytest.mark.parametrize("world", [2, 3, 8])
def test_subtree_partition_medium(gpu, world):
    prob = oa.BAProblem(49, 900, 10000, seed=7)
    res, data = _partition_step(prob, world)
    w = _check_partition_against_oracle(res, data)
    st = res[0]["stats"]
    print(f"world={world}: cut above level {st['cut_level']}, rows/rank {[len(r['rows']) for r in res]}, "
          f"{st['reduced_doubles']*8/1e3:.0f} KB summed per factorisation, |step - oracle| = {w:.2e}")


@pytest.mark.parametrize("world", [2, 8])
def test_subtree_partition_config3_200k_rows(gpu, world):
    """BASELINE.json config #3 (200 000 measurement rows) on logical ranks: every rank's step equals
    the oracle's within the parity bar, and all ranks hold identical bits"""
    prob = oa.BAProblem(499, 9000, 100000, seed=11)
    res, data = _partition_step(prob, world)
    w = _check_partition_against_oracle(res, data)
    st = res[0]["stats"]
    assert st["reduced_doubles"] * 8 < 8e6                     # a few MB, against 16 MB of JtJ
    print(f"config #3, world={world}: cut above level {st['cut_level']}, {st['supernodes_above_cut']} replicated supernodes, "
          f"rows/rank {[len(r['rows']) for r in res]}, {st['reduced_doubles']*8/1e6:.2f} MB summed, |step - oracle| = {w:.2e}")


@pytest.mark.parametrize("world,take", [(2, False), (3, True), (8, True)])
def test_subtree_partition_with_one_pass_evaluation(gpu, world, take):
    """dlg_backend_set_speculation on a partitioned backend: every rank assembles its rows' JtJ in the pass
    that forms its share of Jt*x (sparse_eval_assemble), the sums over the ranks follow where they always
    did -- the oracle's step within the parity bar on every rank, identical bits across ranks"""
    prob = oa.BAProblem(49, 900, 10000, seed=7)
    res, data = _partition_step(prob, world, use_take_step=take, one_pass=True)
    _check_partition_against_oracle(res, data)


def test_sparse_rows_sharded_with_one_pass_evaluation(gpu):
    prob = oa.BAProblem(49, 900, 10000, seed=7)
    N, M, nnz = prob.N, prob.M, prob.nnz
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    cuts = [0, M // 3 + 1, M]
    a = _sharded_step(capi.DLG_SPARSE, N, M, nnz, (Jp, Ji), x, Jx, p, cuts, lambda a, b: Jx[Jp[a]:Jp[b]], one_pass=True)
    b = _sharded_step(capi.DLG_SPARSE, N, M, nnz, (Jp, Ji), x, Jx, p, cuts, lambda a, b: Jx[Jp[a]:Jp[b]], one_pass=False)
    for ra, rb in zip(a, b):
        assert np.linalg.norm(ra["step"] - rb["step"]) <= 1e-11 * max(1.0, np.linalg.norm(rb["step"]))
        assert abs(ra["n2x"] - rb["n2x"]) <= 1e-13 * rb["n2x"]
    assert np.array_equal(a[0]["step"], a[1]["step"])


def test_subtree_partition_take_step_one_synchronisation(gpu):
    """dlg_take_step (K3..K8 behind one host synchronisation) works on a partitioned backend: same
    numbers as the separate calls on a single rank"""
    prob = oa.BAProblem(49, 900, 10000, seed=7)
    res, data = _partition_step(prob, 4, use_take_step=True)
    _check_partition_against_oracle(res, data)


def test_subtree_partition_lambda_loop_agrees_across_ranks(gpu):
    """numerically-zero columns: the non-positive pivot shows up in ONE rank's subtree, but every rank
    must see the failed factorisation and raise lambda together (dogleg.c:656-677)"""
    prob = oa.BAProblem(49, 900, 10000, seed=9, n_zero_cols=2)
    res, data = _partition_step(prob, 4)
    assert all(r["lam"] == 1e-10 for r in res)
    _check_partition_against_oracle(res, data, lam=1e-10, tol=1e-6)


def test_subtree_partition_single_rank_is_the_plain_path(gpu):
    """nranks = 1: the partitioned backend is the ordinary one, bit for bit"""
    prob = oa.BAProblem(12, 120, 720, seed=4)
    res, data = _partition_step(prob, 1)
    N, M, Jp, Ji, x, Jx, p = data
    be = capi.Backend(capi.DLG_SPARSE, N, M, prob.nnz)
    be.set_pattern(Jp, Ji)
    be.set_p(0, p)
    be.upload(0, x, Jx)
    be.eval(0)
    be.cauchy(0)
    lam, n2g = be.gauss_newton(0, 0.0)
    assert np.array_equal(be.download(0, capi.VEC_GN), res[0]["gn"])
    be.close()


# ---------------------------------------------- multi-GPU behind dogleg.h (dogleg_amd_set_allreduce) -------
def _multi_rank_solve(kind, prob, world, prm, twin=None):
    """`world` host threads = logical ranks, each calling the PUBLIC entry point (dogleg_optimize2 /
    _dense2 / _device2) with the same arguments after dogleg_amd_set_allreduce: the driver partitions the
    rows, takes its rank's from what the (all-rows) callback wrote, and sums over the ranks through the hook"""
    L = capi.lib()
    ar = _InProcessAllReduce(world)
    out, errs, hooks = [None] * world, [], []
    p0 = prob.p0()
    N, M = prob.N, prob.M
    nnz = prob.nnz if kind == "sparse" else 0
    pat = prob.pattern() if (kind == "sparse" and twin is not None) else (None, None)

    def run(rank):
        try:
            hook = capi.ALLREDUCE_FN(ar.hook(rank))
            hooks.append(hook)
            assert L.dogleg_amd_set_allreduce(rank, world, -1, C.cast(hook, C.c_void_p), None) == 0
            try:
                if twin is not None:
                    out[rank] = capi.optimize_device(p0, N, M, nnz, pat[0], pat[1], twin.cb, twin.cookie, prm)
                else:
                    cb = prob.cb
                    out[rank] = capi.optimize(kind, p0, N, M, nnz, cb, prob.cookie, prm)
            finally:
                L.dogleg_amd_clear_communicator()
        except Exception as e:
            errs.append((rank, repr(e)))
            try:
                ar.bar.abort()
            except Exception:
                pass

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=900) for t in th]
    assert not errs, errs
    return out


def _check_ranks_against_oracle(kind, prob, prm, res):
    from tests.parity import compare_traces
    nnz = prob.nnz if kind == "sparse" else 0
    ro, po, tro = oa.oracle_solve(kind, prob.p0(), prob.N, prob.M, nnz, prob.cb, prob.cookie, prm)
    worst = 0.0
    for r, p, tr in res:
        assert r >= 0, "a rank's solve failed"
        worst = max(worst, compare_traces(tr, tro))           # every trial: step type, acceptance, |step diff| <= 1e-10
        assert np.max(np.abs(p - po)) <= 1e-10
        assert abs(r - ro) <= 1e-9 * max(1.0, ro)
    for r, p, tr in res[1:]:                                   # every rank: the same bits
        assert r == res[0][0] and np.array_equal(p, res[0][1])
        assert tr.ntrials == res[0][2].ntrials
        for i in range(min(tr.ntrials, tr.capacity)):
            assert np.array_equal(tr.step[i], res[0][2].step[i])
    return worst


@pytest.mark.parametrize("world", [2, 8])
def test_dogleg_optimize2_on_logical_ranks_matches_the_oracle_trace(gpu, world):
    """VERDICT r2, row 8e': multi-GPU reachable behind dogleg_optimize2.  A full sparse solve (Cauchy, GN and
    interpolated steps, accepted and rejected trials) on 2 and 8 logical ranks against the oracle's trace"""
    prob = oa.BAProblem(49, 900, 10000, seed=4, eps=0.4, p0_spread=0.6)
    prm = oa.default_params()
    prm.max_iterations = 12
    prm.trustregion0 = 3.0
    res = _multi_rank_solve("sparse", prob, world, prm)
    w = _check_ranks_against_oracle("sparse", prob, prm, res)
    print(f"dogleg_optimize2 on {world} logical ranks: {res[0][2].ntrials} trials, max |step - oracle| = {w:.2e}")


def test_dogleg_optimize_dense2_on_two_logical_ranks(gpu):
    dp = oa.DenseProblem(M=1201, N=96, seed=2)
    prm = oa.default_params()
    prm.max_iterations = 8
    res = _multi_rank_solve("dense", dp, 2, prm)
    _check_ranks_against_oracle("dense", dp, prm, res)


def test_dogleg_optimize_device2_on_three_logical_ranks(gpu):
    """the model evaluated ON the device for all rows, the rank's rows gathered on the device
    (dlg_point_gather_device): same trace as the oracle's"""
    prob = oa.BAProblem(49, 900, 10000, seed=4, eps=0.4, p0_spread=0.6)
    prm = oa.default_params()
    prm.max_iterations = 8
    prm.trustregion0 = 3.0
    twins = oa.DeviceTwin(prob)
    try:
        res = _multi_rank_solve("sparse", prob, 3, prm, twin=twins)
    finally:
        twins.close()
    _check_ranks_against_oracle("sparse", prob, prm, res)


def test_subtree_partition_config4_on_8_logical_ranks(gpu):
    """BASELINE.json config #4 (1M rows, 150k parameters: the configuration that names 8 GPUs) over 8
    logical ranks on the one device: every rank's step equals the oracle's within the parity bar, identical
    bits across ranks; dlg_take_step with the Cauchy pass beside the factorisation and its scalar summed with
    the solution"""
    prob = oa.BAProblem(2499, 45000, 500000, seed=11)
    res, data = _partition_step(prob, 8, use_take_step=True, one_pass=True)
    w = _check_partition_against_oracle(res, data)
    st = res[0]["stats"]
    assert st["reduced_doubles"] * 8 < 4e6
    print(f"config #4, 8 logical ranks: cut above level {st['cut_level']}, rows/rank {[len(r['rows']) for r in res]}, "
          f"{st['reduced_doubles']*8/1e6:.2f} MB summed per factorisation, |step - oracle| = {w:.2e}")


def test_replicas_of_a_supernode_do_not_race_for_its_panel(gpu):
    """regression (round 4): eight logical ranks run their one-launch regions on ONE device at the same time -- more
    workgroups than CUs, so some replicas of a supernode start late.  Every replica reads the supernode's panel and one
    of them writes the factored panel back: that one used to be the first, and a late replica then read a panel that
    was factored already (a wrong step in 4 runs of 10).  Now the last replica stores, after the others have said that
    their copy is in LDS.  Six rounds, each against the oracle and bit-identical to the first."""
    prob = oa.BAProblem(2499, 45000, 500000, seed=11)
    first = None
    for it in range(6):
        res, data = _partition_step(prob, 8, use_take_step=True, one_pass=True)
        _check_partition_against_oracle(res, data)
        if first is None:
            first = res[0]["step"].copy()
        assert np.array_equal(res[0]["step"], first), it



@pytest.mark.parametrize("shape,world", [((199, 3600, 40000), 4), ((499, 9000, 100000), 8)])
def test_lower_region_of_a_partition_changes_no_bit(gpu, shape, world, monkeypatch):
    """Round 4's second one-launch region -- a rank's own levels from the first multifrontal one up to the cut as ONE
    launch with replicas -- against one launch per level (DOGLEG_AMD_NO_LOWER_REGION): the same step, bit for bit,
    on every rank (INTEGRATION.md says so; VERDICT r4: nothing checked it)."""
    prob = oa.BAProblem(*shape, seed=7)
    monkeypatch.delenv("DOGLEG_AMD_NO_LOWER_REGION", raising=False)
    res, data = _partition_step(prob, world, use_take_step=True, one_pass=True)
    _check_partition_against_oracle(res, data)
    monkeypatch.setenv("DOGLEG_AMD_NO_LOWER_REGION", "1")
    ref, _ = _partition_step(prob, world, use_take_step=True, one_pass=True)
    for a, b in zip(res, ref):
        assert np.array_equal(a["step"], b["step"]) and np.array_equal(a["gn"], b["gn"]) and np.array_equal(a["g"], b["g"])
        assert a["k"] == b["k"] and a["ei"] == b["ei"] and a["n2c"] == b["n2c"] and a["n2g"] == b["n2g"]
