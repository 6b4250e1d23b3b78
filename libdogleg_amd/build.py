"""Build libdogleg_amd.so (HIP kernels + C-ABI + host driver) for gfx950.

hipcc cross-compiles without a GPU.  Objects are cached per source under
libdogleg_amd/csrc/_obj and rebuilt when the source or a header changes.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdogleg_amd.so")
# The reference builds libdogleg.so.2 (Makefile:7, ABI_VERSION := 2).  The library carries that SONAME and
# the two names a linker / loader looks for sit next to it, so `-ldogleg` and binaries that were linked
# against the reference's library resolve to this one once the directory is on the library path.
SONAME = "libdogleg.so.2"
LINKS = ["libdogleg.so.2", "libdogleg.so"]
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def headers():
    inc = os.path.join(os.path.dirname(HERE), "include")
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return hs


# per-source extra flags.  The sparse assembly and factor kernels: their fp64 MFMA loops keep few accumulators; with the
# default heuristic the compiler parks them in AGPRs and copies all of them to VGPRs and back in
# every loop iteration, the VGPR form of the instruction avoids that.
_VGPR_MFMA = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
EXTRA = {"sparse_assemble.hip": _VGPR_MFMA, "sparse_factor.hip": _VGPR_MFMA, "dense_diag.hip": _VGPR_MFMA}


def _compile(src, newest_hdr, verbose):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_hdr):
        return obj
    if src.endswith(".hip"):
        cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    else:       # plain host C++
        cmd = [HIPCC, "-x", "c++"] + [f for f in FLAGS if not f.startswith("--offload-arch")] + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    newest_hdr = max(os.path.getmtime(h) for h in headers())
    srcs = sources()
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, newest_hdr, verbose), srcs))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,-soname,{SONAME}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    ensure_links()
    return LIB


def ensure_links():
    """libdogleg.so.2 and libdogleg.so -> libdogleg_amd.so (relative symlinks, re-made if missing)"""
    for name in LINKS:
        path = os.path.join(HERE, name)
        if os.path.islink(path) and os.readlink(path) == os.path.basename(LIB):
            continue
        if os.path.lexists(path):
            os.remove(path)
        os.symlink(os.path.basename(LIB), path)


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
