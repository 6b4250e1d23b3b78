"""The reference's own test programs against the drop-in (reference check.sh:11-15).

oracle/_ref/sample_dropin and test_misc_dropin are the reference's sample.c and test-misc.c,
compiled unmodified from /root/reference against include/dogleg.h and linked to
libdogleg_amd.so (`make -C oracle dropin`, run by __graft_entry__.build() where the reference is
present; the binaries travel to the GPU box, the sources do not).  check.sh runs
`./sample --check <mode>` for four modes and `./test-misc`; so do these tests."""
import json
import math
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from libdogleg_amd import build as _build
_build.ensure_links()          # libdogleg.so.2 / libdogleg.so next to libdogleg_amd.so (symlinks may not have travelled)
SAMPLE = os.path.join(ROOT, "oracle", "_ref", "sample_dropin")
MISC = os.path.join(ROOT, "oracle", "_ref", "test_misc_dropin")
MODES = ["sparse", "dense", "dense-products-packed-upper", "dense-products-unpacked"]
needs_bins = pytest.mark.skipif(not (os.path.exists(SAMPLE) and os.path.exists(MISC)),
                                reason="oracle/_ref not built (no /root/reference at build time)")


def test_library_carries_the_reference_soname():
    """reference Makefile:7 (ABI_VERSION := 2): the shared library is libdogleg.so.2; ours carries that
    SONAME and the names `-ldogleg` / the loader look for resolve to it"""
    lib = os.path.join(ROOT, "libdogleg_amd", "libdogleg_amd.so")
    r = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True)
    assert "soname: [libdogleg.so.2]" in r.stdout
    for name in ("libdogleg.so.2", "libdogleg.so"):
        assert os.path.realpath(os.path.join(ROOT, "libdogleg_amd", name)) == os.path.realpath(lib)
    if os.path.exists(SAMPLE):
        r = subprocess.run(["readelf", "-d", SAMPLE], capture_output=True, text=True)
        assert "[libdogleg.so.2]" in r.stdout          # the reference's program asks for the reference's library name


@needs_bins
def test_reference_test_misc_passes():
    """check.sh:15 -- the parameter struct's bit layout (host only)"""
    r = subprocess.run([MISC], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DOES match" in r.stdout, r.stdout + r.stderr


@needs_bins
def test_reference_sample_fails_loudly_without_a_gpu():
    """no HIP device: the library has no CPU fallback, the reference's program reports the failure"""
    from libdogleg_amd import capi
    if capi.lib().dlg_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([SAMPLE, "--check", "sparse"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr


@needs_bins
@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_reference_sample_check_passes_on_the_gpu(gpu, mode):
    """check.sh:11-14: `sample --check <mode>` exits 0 iff the recovered parameters are within 5e-2
    of (1..6) (sample.c:424-458)"""
    r = subprocess.run([SAMPLE, "--check", mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@needs_bins
@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_reference_sample_vnlog_stream_matches_the_golden_trace(gpu, mode):
    """`sample --diag vnlog <mode>`: the vnlog records the driver prints for the reference's program are
    those of SURVEY.md Appendix B (tests/golden/sample_trace.json), field by field at %g precision"""
    t = json.load(open(os.path.join(ROOT, "tests", "golden", "sample_trace.json")))
    r = subprocess.run([SAMPLE, "--diag", "vnlog", mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split() for l in r.stdout.splitlines() if l and (l[0].isdigit()) and len(l.split()) == len(t["vnlog_columns"])]
    assert len(rows) == len(t["vnlog"]), r.stdout[-3000:]
    for got, want in zip(rows, t["vnlog"]):
        for g, w in zip(got, want):
            if w is None:
                assert g == "-", (got, want)
            elif isinstance(w, str):
                assert g == w, (got, want)
            else:
                # %g fields; the terminal record's expected improvement is a difference of nearly equal
                # tiny terms (4 digits)
                rel = 1e-4 if (want is t["vnlog"][-1]) else 2e-5
                assert math.isclose(float(g), float(w), rel_tol=rel, abs_tol=1e-300), (got, want)


@needs_bins
@pytest.mark.parametrize("mode", ["sparse", "dense"])
def test_reference_sample_gradient_tables(mode):
    """`sample --test-gradients <mode>` (sample.c:392-405): the reference's program drives
    dogleg_testGradient{,_dense} for every variable and returns; host only.  6 tables of 100 rows,
    reported and observed gradients agree."""
    r = subprocess.run([SAMPLE, "--test-gradients", mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert sum(1 for l in lines if l.startswith("# ivar imeasurement")) == 6
    rows = [l.split() for l in lines if l and l[0].isdigit()]
    assert len(rows) == 600
    for ivar in range(6):
        sub = [x for x in rows if int(x[0]) == ivar]
        assert [int(x[1]) for x in sub] == list(range(100))
        rep = [float(x[2]) for x in sub]
        obs = [float(x[3]) for x in sub]
        scale = max(1.0, max(abs(v) for v in rep))
        assert max(abs(a - b) for a, b in zip(rep, obs)) <= 1e-4 * scale
