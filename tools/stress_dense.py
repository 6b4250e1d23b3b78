#!/usr/bin/env python3
"""Random sizes through the dense ops (K1, K3, K4+K5, K6) against numpy: odd N and M around the tile
sizes of the SYRK / potrf / trsv kernels.  usage: stress_dense.py [n] [seed0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdogleg_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for s in range(seed0, seed0 + n):
    rng = np.random.default_rng(s)
    N = int(rng.choice([1, 2, 3, 7, 8, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128, 129, 191, 200, 255, 257, 300, 511, 513, 700]))
    M = int(N + rng.integers(0, 5 * N + 40))
    J = rng.standard_normal((M, N))
    x = rng.standard_normal(M)
    p = rng.standard_normal(N)
    try:
        be = capi.Backend(capi.DLG_DENSE, N, M)
        be.set_p(0, p)
        be.upload(0, x, J.reshape(-1))
        n2x, gmax = be.eval(0)
        g = J.T @ x
        assert abs(n2x - x @ x) <= 1e-12 * (x @ x)
        assert np.max(np.abs(be.download(0, capi.VEC_JTX) - g)) <= 1e-11 * max(1.0, np.max(np.abs(g)))
        n2c = be.cauchy(0)
        k = -(g @ g) / ((J @ g) @ (J @ g))
        assert abs(n2c - k * k * (g @ g)) <= 1e-10 * max(n2c, 1e-300)
        lam, n2g = be.gauss_newton(0, 0.0)
        A = J.T @ J + lam * np.eye(N)
        ref = -np.linalg.solve(A, g)
        gn = be.download(0, capi.VEC_GN)
        cond = np.linalg.cond(A)
        err = np.linalg.norm(gn - ref) / max(1.0, np.linalg.norm(ref))
        assert err <= max(1e-10, 100 * cond * 1.1e-16), (err, cond)
        be.close()
    except Exception as e:
        bad += 1
        print("FAIL seed", s, "N", N, "M", M, repr(e)[:300], flush=True)
print(f"{n - bad} of {n} passed")
sys.exit(1 if bad else 0)
