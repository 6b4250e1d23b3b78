"""GPU parity on sparsity patterns that are NOT bundle-adjustment shaped: the symbolic phase
and the kernels are generic; these exercise single-variable blocks, long rows (unstaged
assembly path), dense-as-sparse, banded chains, empty rows and never-touched variables."""
import ctypes as C
import numpy as np
import pytest

from libdogleg_amd import capi
from libdogleg_amd.ctypes_defs import dptr, iptr
from tests import oracle_api as oa

pytestmark = pytest.mark.gpu


def _check_pattern(N, M, Jp, Ji, Jx, x, lam=0.0, tol=1e-9):
    O = oa.oracle()
    Jp = np.ascontiguousarray(Jp, dtype=np.int32)
    Ji = np.ascontiguousarray(Ji, dtype=np.int32)
    nnz = int(Jp[-1])
    be = capi.Backend(capi.DLG_SPARSE, N, M, nnz)
    be.set_pattern(Jp, Ji)
    be.set_p(0, np.zeros(N))
    be.upload(0, x, Jx)
    n2x, gmax = be.eval(0)
    g_ref = np.zeros(N)
    O.orc_spmv_Jt_x(dptr(g_ref), N, M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(x))
    g = be.download(0, capi.VEC_JTX)
    assert np.max(np.abs(g - g_ref)) <= 1e-12 * max(1.0, np.max(np.abs(g_ref)))
    n2c = be.cauchy(0)
    Jg2 = O.orc_norm2_J_v(M, iptr(Jp), iptr(Ji), dptr(Jx), dptr(g_ref))
    k = -(g_ref @ g_ref) / Jg2
    assert abs(n2c - k * k * (g_ref @ g_ref)) <= 1e-10 * max(1e-300, n2c)
    ok = be.factorize(0, lam)
    F = O.orc_sparse_analyze(N, M, iptr(Jp), iptr(Ji))
    ok_ref = O.orc_sparse_factorize(F, iptr(Jp), iptr(Ji), dptr(Jx), lam) == N
    assert ok == ok_ref
    if ok:
        be.solve_gn(0)
        ref = np.zeros(N)
        O.orc_sparse_solve(F, dptr(g_ref), dptr(ref))
        gn = be.download(0, capi.VEC_GN)
        err = np.linalg.norm(gn + ref) / max(1.0, np.linalg.norm(ref))
        assert err <= tol, err
    O.orc_sparse_free(F)
    st = be.stats()
    be.close()
    return st


def _rows_to_csc(rows, N):
    Jp = [0]
    Ji = []
    for r in rows:
        r = sorted(set(int(i) for i in r))
        Ji.extend(r)
        Jp.append(len(Ji))
    return np.array(Jp, dtype=np.int32), np.array(Ji, dtype=np.int32)


def test_random_sparse_pattern(gpu):
    rng = np.random.default_rng(3)
    N, M = 400, 2500
    rows = [rng.choice(N, size=rng.integers(2, 7), replace=False) for _ in range(M)]
    Jp, Ji = _rows_to_csc(rows, N)
    Jx = rng.standard_normal(Jp[-1])
    x = rng.standard_normal(M)
    st = _check_pattern(N, M, Jp, Ji, Jx, x)
    print("random sparse:", st)


def test_dense_rows_given_as_sparse(gpu):
    """every row touches every variable (the reference's own sample problem is like this);
    rows of 300 entries exceed the LDS staging budget -> unstaged assembly path"""
    rng = np.random.default_rng(4)
    N, M = 300, 700
    Jp = np.arange(0, (M + 1) * N, N, dtype=np.int32)
    Ji = np.tile(np.arange(N, dtype=np.int32), M)
    Jx = rng.standard_normal(M * N)
    x = rng.standard_normal(M)
    st = _check_pattern(N, M, Jp, Ji, Jx, x, tol=1e-8)
    print("dense-as-sparse:", st)


def test_banded_chain_pattern(gpu):
    """tridiagonal-like JtJ: a long elimination chain (exercises nested dissection leaves,
    chain supernodes, many levels)"""
    rng = np.random.default_rng(5)
    N = 3000
    rows = [[i, i + 1, i + 2] for i in range(N - 2)] + [[i] for i in range(0, N, 7)]
    Jp, Ji = _rows_to_csc(rows, N)
    M = len(rows)
    Jx = rng.standard_normal(Jp[-1]) + 2.0
    x = rng.standard_normal(M)
    st = _check_pattern(N, M, Jp, Ji, Jx, x, tol=1e-7)
    print("banded:", st)


def test_empty_rows_and_untouched_variables(gpu):
    """measurement rows without entries and state variables no row touches: JtJ has zero
    rows/columns -> not positive definite at lambda = 0, fine once lambda > 0
    (reference semantics: dogleg.c:656-677)"""
    rng = np.random.default_rng(6)
    N, M = 60, 200
    rows = []
    for r in range(M):
        if r % 17 == 0:
            rows.append([])
        else:
            rows.append(rng.choice(np.arange(0, N - 5), size=4, replace=False))   # last 5 vars untouched
    Jp, Ji = _rows_to_csc(rows, N)
    Jx = rng.standard_normal(Jp[-1])
    x = rng.standard_normal(M)
    _check_pattern(N, M, Jp, Ji, Jx, x, lam=0.0)          # both sides must report "singular"
    _check_pattern(N, M, Jp, Ji, Jx, x, lam=1e-6, tol=1e-6)


def test_block_sizes_around_the_limits(gpu):
    """variable blocks of 1..20 consecutive variables (blocks > 8 are split) and row-blocks of
    up to 12 identical rows (> 8 are split)"""
    rng = np.random.default_rng(7)
    sizes = [1, 2, 3, 5, 8, 9, 13, 20, 4, 7]
    starts = np.concatenate([[0], np.cumsum(sizes)])
    N = int(starts[-1])
    rows = []
    for _ in range(120):
        pick = rng.choice(len(sizes), size=3, replace=False)
        idx = np.concatenate([np.arange(starts[b], starts[b + 1]) for b in pick])
        for _ in range(int(rng.integers(1, 13))):
            rows.append(idx)
    Jp, Ji = _rows_to_csc(rows, N)
    M = len(rows)
    Jx = rng.standard_normal(Jp[-1])
    x = rng.standard_normal(M)
    st = _check_pattern(N, M, Jp, Ji, Jx, x, tol=1e-8)
    print("block limits:", st)


def test_wide_dense_tail(gpu):
    """2560 variables that every row touches (rows of 320 variable blocks): target slabs too big
    for LDS (HBM-accumulating update path), panels cut into many row slices, a long chain of
    supernodes, the LDS assembly kernel with hundreds of blocks per row"""
    rng = np.random.default_rng(8)
    N, M = 2560, 2700
    Jp = np.arange(0, (M + 1) * N, N, dtype=np.int64).astype(np.int32)
    Ji = np.tile(np.arange(N, dtype=np.int32), M)
    Jx = rng.standard_normal(M * N)
    Jx[::N + 1] += 30.0          # make JtJ comfortably positive definite
    x = rng.standard_normal(M)
    st = _check_pattern(N, M, Jp, Ji, Jx, x, tol=1e-7)
    print("wide dense tail:", st)


def _random_structured_pattern(rng, scale=1):
    """variable blocks of random widths, a few of them 'dense' (touched by every row), rows drawn
    from a small set of templates and repeated 1..10 times (row-blocks), random values"""
    nblk = int(rng.integers(6, 40))*scale
    widths = rng.integers(1, 10, size=nblk)
    starts = np.concatenate([[0], np.cumsum(widths)])
    N = int(starts[-1])
    ndense = int(rng.integers(0, 3))
    dense_blocks = list(rng.choice(nblk, size=ndense, replace=False)) if ndense else []
    templates = []
    for _ in range(int(rng.integers(3, 60))*scale):
        k = int(rng.integers(1, min(5, nblk) + 1))
        blks = set(rng.choice(nblk, size=k, replace=False).tolist()) | set(dense_blocks)
        templates.append(sorted(blks))
    rows = []
    target = int(rng.integers(N + 20, 6 * N + 200))
    while len(rows) < target:
        t = templates[int(rng.integers(0, len(templates)))]
        idx = np.concatenate([np.arange(starts[b], starts[b + 1]) for b in t])
        for _ in range(int(rng.integers(1, 11))):
            rows.append(idx)
    # every block gets as many rows of its own as it is wide, so that J has full column rank
    # (the values of these rows are boosted by the caller)
    ntail = 0
    for b in range(nblk):
        for _ in range(int(widths[b])):
            rows.append(np.arange(starts[b], starts[b + 1]))
            ntail += int(widths[b])
    return N, rows, ntail


@pytest.mark.parametrize("seed", range(24))
def test_random_structured_patterns(gpu, seed, monkeypatch):
    """stress of the symbolic phase and the schedules (layout classes, riders with a low threshold,
    partial-sum stages, slices, every update path) on randomly structured problems"""
    rng = np.random.default_rng(1000 + seed)
    if seed % 3 == 0:
        monkeypatch.setenv("DOGLEG_AMD_RIDER_MIN", "16")       # riders even on small problems
    if seed % 4 == 1:
        monkeypatch.setenv("DOGLEG_AMD_SYRK_MIN", "2")         # two-phase updates at small levels
    if seed % 5 == 2:
        monkeypatch.setenv("DOGLEG_AMD_SLICE_CAP", "3000")     # sliced panels
    N, rows, ntail = _random_structured_pattern(rng)
    Jp, Ji = _rows_to_csc(rows, N)
    M = len(rows)
    Jx = rng.standard_normal(Jp[-1])
    tail = Jp[-1] - ntail
    Jx[tail:] *= 4.0                                           # JtJ comfortably positive definite
    x = rng.standard_normal(M)
    _check_pattern(N, M, Jp, Ji, Jx, x, tol=1e-8)


@pytest.mark.parametrize("seed", [9026, 9030, 9052])
def test_dense_block_with_many_row_layouts(gpu, seed, monkeypatch):
    """regression: a dense column block (no tasks of its own, carried by other column blocks) whose
    left-over rows come in more than 64 different layouts used to have no assembly schedule"""
    monkeypatch.setenv("DOGLEG_AMD_RIDER_MIN", "8")
    rng = np.random.default_rng(seed)
    N, rows, ntail = _random_structured_pattern(rng, scale=20)
    Jp, Ji = _rows_to_csc(rows, N)
    M = len(rows)
    Jx = rng.standard_normal(Jp[-1])
    Jx[Jp[-1] - ntail:] *= 4.0
    x = rng.standard_normal(M)
    _check_pattern(N, M, Jp, Ji, Jx, x, tol=1e-6)


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_ragged_rows_with_every_alignment_of_the_value_runs(gpu, seed):
    """K3/K8 read four non-zeros per 16-byte load from the 4-element boundary below a run's first value:
    rows of 1..9 entries put the runs of the workgroups at every offset, the total count at every
    remainder (the last thread goes element by element past the end of the arrays)"""
    rng = np.random.default_rng(seed)
    N = 120
    M = 3000 + seed          # several runs of <= 2048 values
    rows = [rng.choice(N, size=rng.integers(1, 10), replace=False) for _ in range(M)]
    rows[-1] = rows[-1][:1 + seed % 3]
    Jp, Ji = _rows_to_csc(rows, N)
    Jx = rng.standard_normal(Jp[-1])
    x = rng.standard_normal(M)
    _check_pattern(N, M, Jp, Ji, Jx, x)
