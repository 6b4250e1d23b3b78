"""An independent leg for pinning the oracle (and the GPU path) at the 1e-10 level.

The reference delegates its factor / solve arithmetic to third-party libraries: LAPACK's
dpptrf/dpptrs (dogleg.c:782,875) and dpotrf/dpotrs (801,889) on the dense paths, CHOLMOD on the
sparse path (659-664, 853).  The real LAPACK routines are in this image (scipy.linalg.lapack, the
same Fortran entry points the reference links), and scipy's SuperLU (scipy.sparse.linalg.splu) is
an exact sparse direct solver that shares no code with the oracle or with the HIP kernels.  This
module re-derives ONE trial step of takeStepFrom (dogleg.c:1172-1297) from the problem's x and J
with those libraries and numpy only:

    g = Jt x;  cauchy = -(|g|^2 / |J g|^2) g           (dogleg.c:529-617)
    gn = -(JtJ + lambda I)^-1 g                         (dogleg.c:822-908; LAPACK / SuperLU)
    the choice of step and the interpolation           (dogleg.c:1192-1256, 964-987)

Nothing here imports the oracle or the product: tests compare both against it.
"""
import numpy as np
import scipy.linalg.lapack as lapack
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def gn_dense_lapack(J, g, lam=0.0, packed=True):
    """-(JtJ + lam I)^-1 g with the LAPACK routines the reference calls.  Returns (gn, info)."""
    N = J.shape[1]
    A = J.T @ J
    A[np.diag_indices(N)] += lam
    if packed:
        # dpptrf_('L') on column-major packed lower == the reference's row-major packed upper
        ap = np.concatenate([A[j:, j] for j in range(N)])        # column-major lower, column by column
        c, info = lapack.dpptrf(N, ap, lower=1)
        if info != 0:
            return None, info
        sol, info2 = lapack.dpptrs(N, c, g.copy(), lower=1)
        assert info2 == 0
        return -sol, 0
    c, info = lapack.dpotrf(A, lower=1)
    if info != 0:
        return None, info
    sol, info2 = lapack.dpotrs(c, g.copy(), lower=1)
    assert info2 == 0
    return -sol, 0


def gn_sparse_splu(Jcsr, g, lam=0.0, refine=2, perm=None):
    """-(JtJ + lam I)^-1 g with SuperLU, then `refine` rounds of iterative refinement with the
    residual accumulated in extended precision (np.longdouble): the result is accurate far below
    the 1e-10 parity bar, whatever the conditioning of the fixtures in use.
    perm (optional): a fill-reducing elimination order supplied by the caller -- position k eliminates variable
    perm[k]; SuperLU then factors P A P' in its NATURAL order.  (A permutation is not arithmetic: config #5's JtJ,
    500 001 x 500 001 with 97.6 M non-zeros, runs SuperLU's own minimum-degree ordering out of memory in this
    container; with the nested-dissection order of the product's symbolic phase nnz(L) is 60 M.)"""
    N = Jcsr.shape[1]
    A = (Jcsr.T @ Jcsr).tocsc()
    if lam:
        A = A + lam * sp.identity(N, format="csc")
    if perm is None:
        lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                       options=dict(SymmetricMode=True))
        solve = lu.solve
    else:
        perm = np.asarray(perm, dtype=np.int64)
        Ap = A[perm][:, perm].tocsc()
        lu = spla.splu(Ap, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
        del Ap

        def solve(b):
            out = np.empty_like(b)
            out[perm] = lu.solve(np.ascontiguousarray(b[perm]))
            return out
    u = solve(g)
    if refine:
        Ac = A.tocoo()
        r_, c_, v_ = Ac.row, Ac.col, Ac.data.astype(np.longdouble)
        for _ in range(refine):
            res = g.astype(np.longdouble).copy()
            np.subtract.at(res, r_, v_ * u.astype(np.longdouble)[c_])
            u = u + solve(res.astype(np.float64))
    return -u


def gn_ba_schur(Jcsr, g, lam, n_lead, blk, refine=8, tol=1e-14, log=None):
    """-(JtJ + lam I)^-1 g for a block-arrowhead J whose last columns are independent blocks of `blk` columns (no
    measurement row touches two of them: the points of a bundle adjustment; the first n_lead columns are the rest):
    block elimination -- the reduced system  S = Jc'Jc + lam I - E' (D + lam I)^-1 E  (E = Jp'Jc, D = Jp'Jp, block
    diagonal) factored by SuperLU, the blocks by numpy's batched LAPACK inverse -- as the INNER solver of an iterative
    refinement whose residual  g - (J'(J u) + lam u)  is accumulated in extended precision (np.longdouble).  The
    fixed point of the refinement is the solution of the system itself, whatever the inner solver's rounding; the
    loop runs until the correction is below tol |u| (the caller records the last corrections and the residual).
    For config #5's JtJ (500 001 x 500 001, 97.6 M non-zeros) scipy's SuperLU gives up on the whole matrix in this
    container (32-bit fill estimate / memory), on the reduced system (50 004 x 50 004) it does not."""
    M, N = Jcsr.shape
    nb = (N - n_lead) // blk
    assert n_lead + nb * blk == N
    Jc = Jcsr[:, :n_lead].tocsr()
    Jp = Jcsr[:, n_lead:].tocsr()
    D = (Jp.T @ Jp).tobsr(blocksize=(blk, blk))
    D.sort_indices()
    assert D.nnz == nb * blk * blk and np.array_equal(D.indices, np.arange(nb)), "the trailing blocks are coupled"
    Dinv = np.linalg.inv(D.data + lam * np.eye(blk)[None, :, :])
    Dinv = sp.bsr_matrix((Dinv, np.arange(nb), np.arange(nb + 1)), shape=(nb * blk, nb * blk)).tocsr()
    E = (Jp.T @ Jc).tocsr()
    S = (Jc.T @ Jc) + lam * sp.identity(n_lead, format="csr") - E.T @ (Dinv @ E)
    lu = spla.splu(S.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    del S

    def inner(b):
        bp = Dinv @ b[n_lead:]
        yc = lu.solve(b[:n_lead] - E.T @ bp)
        return np.concatenate([yc, bp - Dinv @ (E @ yc)])
    Jcoo = Jcsr.tocoo()
    rr, cc, vv = Jcoo.row, Jcoo.col, Jcoo.data.astype(np.longdouble)

    def residual(u):
        ul = u.astype(np.longdouble)
        y = np.zeros(M, dtype=np.longdouble)
        np.add.at(y, rr, vv * ul[cc])
        res = g.astype(np.longdouble) - np.longdouble(lam) * ul
        np.subtract.at(res, cc, vv * y[rr])
        return res
    u = inner(g)
    hist = []
    for it in range(refine):
        res = residual(u)
        du = inner(res.astype(np.float64))
        u = u + du
        hist.append((float(np.linalg.norm(du) / np.linalg.norm(u)), float(np.sqrt(float(res @ res)) / np.linalg.norm(g))))
        if log:
            log(f"refinement {it}: |du|/|u| = {hist[-1][0]:.3e}, |res|/|g| = {hist[-1][1]:.3e}")
        if hist[-1][0] < tol:
            break
    return -u, hist, residual(u)


def trial_step(J, x, trustregion, lam=0.0, dense_packed=True, gn=None):
    """One step of takeStepFrom from a fresh point.  J: dense ndarray or scipy.sparse matrix.
    Returns dict(step, kind, norm2_cauchy, norm2_gn, k, expected_improvement, g).  kind: 0 Cauchy to
    the edge, 1 Gauss-Newton, 2 interpolated."""
    sparse = sp.issparse(J)
    g = np.asarray(J.T @ x).ravel()
    Jg = np.asarray(J @ g).ravel()
    g2 = float(g @ g)
    kc = -g2 / float(Jg @ Jg)                                   # dogleg.c:605
    cauchy = kc * g
    n2c = kc * kc * g2                                          # dogleg.c:607
    dsq = trustregion * trustregion
    out = dict(g=g, norm2_cauchy=n2c, norm2_gn=np.nan, k=np.nan)
    if n2c >= dsq:                                              # dogleg.c:1192
        step = cauchy * (trustregion / np.sqrt(n2c))
        kind = 0
    else:
        if gn is None:         # (a caller that has the Gauss-Newton step already -- config #5's fixture -- passes it in)
            gn = gn_sparse_splu(J.tocsr(), g, lam) if sparse else gn_dense_lapack(J, g, lam, dense_packed)[0]
        n2g = float(gn @ gn)
        out["norm2_gn"] = n2g
        if n2g <= dsq:                                          # dogleg.c:1220
            step, kind = gn, 1
        else:                                                   # dogleg.c:964-987
            a, b = cauchy, gn
            d = a - b
            l2 = float(d @ d)
            neg_c = float(d @ a)
            disc = neg_c * neg_c - l2 * (n2c - dsq)
            disc = max(disc, 0.0)
            k = (neg_c + np.sqrt(disc)) / l2
            step, kind = a + k * (b - a), 2
            out["k"] = k
    Js = np.asarray(J @ step).ravel()
    out.update(step=step, kind=kind, expected_improvement=-2.0 * float(g @ step) - float(Js @ Js))
    return out


def csr_from_pattern(M, N, Jp, Ji, Jx):
    """rows of J from the CSC arrays of Jt (column r of Jt = row r of J)"""
    return sp.csr_matrix((Jx, Ji, Jp), shape=(M, N))
