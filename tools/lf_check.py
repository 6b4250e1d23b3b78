"""Leaf fronts (sparse_leaf.hip) against the separate assembly / leaf kernels, op by op (GPU, tools only).
usage: python3 tools/lf_check.py [tiny|200k|1m] [spec]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("DOGLEG_AMD_SYRK_MIN", "1")
from libdogleg_amd import capi
import problems

which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
spec = len(sys.argv) > 2 and sys.argv[2] == "spec"
kw = {"tiny": dict(Nc=49, Np=900, Nobs=10000), "200k": dict(Nc=499, Np=9000, Nobs=100000),
      "1m": dict(Nc=2499, Np=45000, Nobs=500000)}[which]
P = problems.BAProblem(**kw)
Jp, Ji = P.pattern()
p = P.p0()
x, Jx = P.eval(p)
print("leaf fronts:", {k: v for k, v in capi.symbolic_probe(P.N, P.M, Jp, Ji).items() if k.startswith("lf") or k == "leaf_fronts"})

def run(off):
    if off:
        os.environ.pop("DOGLEG_AMD_LEAF_FRONT", None)
    else:
        os.environ["DOGLEG_AMD_LEAF_FRONT"] = "1"
    be = capi.Backend(capi.DLG_SPARSE, P.N, P.M, P.nnz)
    be.set_pattern(Jp, Ji)
    if spec:
        be.set_speculation(True)
    be.set_p(0, p)
    be.upload(0, x, Jx)
    n2x, amax = be.eval(0)
    g = be.download(0, capi.VEC_JTX)
    ok = be.factorize(0, 0.0)
    n2gn = be.solve_gn(0)
    gn = be.download(0, capi.VEC_GN)
    be.close()
    return dict(n2x=n2x, amax=amax, g=g, ok=ok, n2gn=n2gn, gn=gn)

a = run(True)
b = run(False)
print("ok", a["ok"], b["ok"], "n2x", a["n2x"], b["n2x"], "amax", a["amax"], b["amax"])
print("Jtx  max|diff| %.3e (scale %.3e)" % (np.max(np.abs(a["g"] - b["g"])), np.max(np.abs(a["g"]))))
print("GN   |diff|/|gn| %.3e   n2gn %.15e %.15e" % (np.linalg.norm(a["gn"] - b["gn"])/np.linalg.norm(a["gn"]), a["n2gn"], b["n2gn"]))
bad = np.argsort(-np.abs(a["gn"] - b["gn"]))[:8]
print("worst entries", bad, a["gn"][bad], b["gn"][bad])
