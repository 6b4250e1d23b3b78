#!/usr/bin/env python3
"""Generates tests/golden/splu_config3_step.json: one full trial step of BASELINE.json
config #3 (200k measurements x 30k parameters, 3M non-zeros; problems.c generator, seed 11)
computed WITHOUT the oracle and without the product -- numpy + scipy's SuperLU with extended
precision iterative refinement (tests/independent.py).  Both the oracle (CPU test) and the HIP
path (-m gpu test) are compared with it at the 1e-10 parity bar: an independent pin of the sparse
factor / solve arithmetic that the reference delegates to CHOLMOD.

The same for config #4 (1M x 150k, 15M non-zeros) -> splu_config4_step.json, every 16th entry of the
vectors plus their norms and sums, to keep the file small.

Data only: hex floats of the Gauss-Newton step, the Cauchy scalars and the interpolated step.
Run here (scipy 1.15.3):  python tests/golden/make_independent_goldens.py [2] [3] [4] [5]   (config #3: about a minute)\nConfig #2 (dense 50 000 x 2 000) -> lapack_config2_step.json: BLAS J'J + the image's LAPACK dpptrf / dpptrs.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy
from tests import oracle_api as oa          # only for the PROBLEM generator (problems.c); no oracle call
from tests import independent as ind

HERE = os.path.dirname(os.path.abspath(__file__))


def hexlist(a):
    return [float(v).hex() for v in np.asarray(a).ravel()]


def make(args, fname, stride):
    """stride > 1: every stride-th entry of the vectors (+ their norms): a small file for a large problem"""
    prob = oa.BAProblem(args["Nc"], args["Np"], args["Nobs"], seed=args["seed"])
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    J = ind.csr_from_pattern(prob.M, prob.N, Jp, Ji, Jx)
    g = np.asarray(J.T @ x).ravel()
    gn = ind.gn_sparse_splu(J, g, 0.0, refine=3)
    # the same trust region bench.py / orc_step_sparse use: mean(|cauchy|, |gn|) -> interpolation
    Jg = np.asarray(J @ g).ravel()
    kc = -float(g @ g) / float(Jg @ Jg)
    n2c = kc * kc * float(g @ g)
    tr = 0.5 * (np.sqrt(n2c) + np.sqrt(float(gn @ gn)))
    st = ind.trial_step(J, x, tr, 0.0)
    # residual of the refined solve, in extended precision: how exact the fixture is
    A = (J.T @ J).tocoo()
    res = g.astype(np.longdouble).copy()
    np.add.at(res, A.row, A.data.astype(np.longdouble) * gn.astype(np.longdouble)[A.col])
    out = {
        "_generator": "tests/golden/make_independent_goldens.py: numpy %s + scipy %s SuperLU (MMD_AT_PLUS_A, symmetric mode) "
                      "+ 3 rounds of long-double iterative refinement; problem = problems.c BAProblem(%d,%d,%d,seed=%d) at p0"
                      % (np.__version__, scipy.__version__, args["Nc"], args["Np"], args["Nobs"], args["seed"]),
        "problem": args, "N": prob.N, "M": prob.M, "nnz": prob.nnz,
        "norm2_x": float(x @ x).hex(),
        "norm2_cauchy": float(n2c).hex(),
        "norm2_gn": float(gn @ gn).hex(),
        "trustregion": float(tr).hex(),
        "kind": int(st["kind"]), "k": float(st["k"]).hex(),
        "expected_improvement": float(st["expected_improvement"]).hex(),
        "relative_residual_of_gn": float(np.sqrt(float(res @ res)) / np.linalg.norm(g)),
        "stride": stride,
        "norm2_step": float(st["step"] @ st["step"]).hex(),
        "sum_gn": float(np.sum(gn)).hex(), "sum_step": float(np.sum(st["step"])).hex(),
        "gn_hex": hexlist(gn[::stride]),
        "step_hex": hexlist(st["step"][::stride]),
    }
    json.dump(out, open(os.path.join(HERE, fname), "w"), indent=0)
    print(fname, "written; relative residual of the refined GN solve:", out["relative_residual_of_gn"])


def make5(args, fname, stride):
    """BASELINE.json config #5 (5M x 500 001, 75M non-zeros, column scales over 4 decades, exactly-zero columns =>
    lambda = 1e-10, dogleg.c:656-677).  scipy's SuperLU gives up on its JtJ whatever the ordering (tried: its own
    minimum-degree ordering -- round 4 --, and NATURAL on the matrix permuted by the product's nested dissection --
    round 5: "Not enough memory to perform factorization" within a minute, its 32-bit fill estimate).  The fixture
    therefore comes from block elimination of the points (tests/independent.py, gn_ba_schur: batched LAPACK inverses of
    the 3 x 3 blocks, SuperLU on the 50 004 x 50 004 reduced system) as the inner solver of an iterative refinement
    with long-double residuals taken through J itself -- no oracle, no product, and the fixed point of the refinement
    does not depend on how the inner solver rounds."""
    prob = oa.BAProblem(args["Nc"], args["Np"], args["Nobs"], seed=args["seed"], scale_decades=args["scale_decades"],
                        n_zero_cols=args["n_zero_cols"])
    lam = args["lambda"]
    Jp, Ji = prob.pattern()
    p = prob.p0()
    x, Jx = prob.eval(p)
    J = ind.csr_from_pattern(prob.M, prob.N, Jp, Ji, Jx)
    g = np.asarray(J.T @ x).ravel()
    n_lead = prob.N - 3 * args["Np"]
    gn, hist, res = ind.gn_ba_schur(J, g, lam, n_lead, 3, refine=12, tol=3e-16, log=print)
    Jg = np.asarray(J @ g).ravel()
    kc = -float(g @ g) / float(Jg @ Jg)
    n2c = kc * kc * float(g @ g)
    tr = 0.5 * (np.sqrt(n2c) + np.sqrt(float(gn @ gn)))
    st = ind.trial_step(J, x, tr, lam, gn=gn)
    out = {
        "_generator": "tests/golden/make_independent_goldens.py: numpy %s + scipy %s; points eliminated in blocks (batched LAPACK inverse), "
                      "SuperLU (MMD_AT_PLUS_A, symmetric mode) on the reduced system, iterative refinement with long-double residuals through J "
                      "until the correction stalls; problem = problems.c BAProblem(%d,%d,%d,seed=%d,scale_decades=%g,n_zero_cols=%d) at p0, lambda = %g"
                      % (np.__version__, scipy.__version__, args["Nc"], args["Np"], args["Nobs"], args["seed"], args["scale_decades"],
                         args["n_zero_cols"], lam),
        "problem": args, "N": prob.N, "M": prob.M, "nnz": prob.nnz, "lambda": float(lam).hex(),
        "norm2_x": float(x @ x).hex(), "norm2_cauchy": float(n2c).hex(), "norm2_gn": float(gn @ gn).hex(),
        "trustregion": float(tr).hex(), "kind": int(st["kind"]), "k": float(st["k"]).hex(),
        "expected_improvement": float(st["expected_improvement"]).hex(),
        "refinement_history_relative_correction_and_residual": hist,
        "relative_residual_of_gn": float(np.sqrt(float(res @ res)) / np.linalg.norm(g)),
        "stride": stride, "norm2_step": float(st["step"] @ st["step"]).hex(),
        "sum_gn": float(np.sum(gn)).hex(), "sum_step": float(np.sum(st["step"])).hex(),
        "gn_hex": hexlist(gn[::stride]), "step_hex": hexlist(st["step"][::stride]),
    }
    json.dump(out, open(os.path.join(HERE, fname), "w"), indent=0)
    print(fname, "written; relative residual of the refined GN solve:", out["relative_residual_of_gn"])


def make_dense(args, fname):
    """BASELINE.json config #2 (dense 50 000 x 2 000): JtJ by BLAS, dpptrf_ / dpptrs_ of the image's LAPACK (the entry points
    the reference links, dogleg.c:782,875) through tests/independent.py -- no oracle, no product"""
    dp = oa.DenseProblem(M=args["M"], N=args["N"], seed=args["seed"])
    p = dp.p0()
    x, J = dp.eval(p)
    g = J.T @ x
    gn, info = ind.gn_dense_lapack(J, g, 0.0, packed=True)
    assert info == 0
    Jg = J @ g
    kc = -float(g @ g) / float(Jg @ Jg)
    n2c = kc * kc * float(g @ g)
    tr = 0.5 * (np.sqrt(n2c) + np.sqrt(float(gn @ gn)))
    st = ind.trial_step(J, x, tr, 0.0)
    A = J.T @ J
    res = np.linalg.norm(A.astype(np.longdouble) @ gn.astype(np.longdouble) + g.astype(np.longdouble))
    out = {
        "_generator": "tests/golden/make_independent_goldens.py: numpy %s (BLAS J'J) + scipy %s LAPACK dpptrf/dpptrs; "
                      "problem = problems.c DenseProblem(M=%d, N=%d, seed=%d) at p0" % (np.__version__, scipy.__version__, args["M"], args["N"], args["seed"]),
        "problem": args, "norm2_x": float(x @ x).hex(), "norm2_cauchy": float(n2c).hex(), "norm2_gn": float(gn @ gn).hex(),
        "trustregion": float(tr).hex(), "kind": int(st["kind"]), "k": float(st["k"]).hex(),
        "expected_improvement": float(st["expected_improvement"]).hex(),
        "relative_residual_of_gn": float(res / np.linalg.norm(g)),
        "gn_hex": hexlist(gn), "step_hex": hexlist(st["step"]),
    }
    json.dump(out, open(os.path.join(HERE, fname), "w"), indent=0)
    print(fname, "written; relative residual of the GN solve:", out["relative_residual_of_gn"])


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4"]
    if "2" in which:
        make_dense(dict(M=50000, N=2000, seed=2), "lapack_config2_step.json")
    if "3" in which:
        make(dict(Nc=499, Np=9000, Nobs=100000, seed=11), "splu_config3_step.json", 1)
    if "4" in which:
        # BASELINE.json config #4 (1M x 150k, 15M nnz): every 16th entry of the vectors + norms and sums
        make(dict(Nc=2499, Np=45000, Nobs=500000, seed=11), "splu_config4_step.json", 16)
    if "5" in which:
        # BASELINE.json config #5 (5M x 500 001, 75M nnz, lambda = 1e-10): every 64th entry + norms and sums (about 10 minutes, ~30 GB)
        make5(dict(Nc=8333, Np=149999, Nobs=2500000, seed=13, scale_decades=4.0, n_zero_cols=3, **{"lambda": 1e-10}), "splu_config5_step.json", 64)
