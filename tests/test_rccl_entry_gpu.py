"""VERDICT r3, row 8e': the RCCL leg of the PUBLIC multi-GPU entry (reference interface it sits behind:
dogleg.h:278-302) executed -- dogleg_amd_set_communicator, and the environment contract with its id file --
at world size 1, the only size a one-GPU box can run (RCCL refuses two ranks on one device).  Every solve is a
full dogleg_optimize2 / _dense2 / _device2 against the oracle's trace; tests/rccl_child.py does the work in a
process of its own."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(how, kind, env_extra=None):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in list(env):
        if k.startswith("DOGLEG_AMD_WORLD") or k in ("DOGLEG_AMD_RANK", "DOGLEG_AMD_RCCL_ID_FILE", "DOGLEG_AMD_FORCE_COMM"):
            del env[k]
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-m", "tests.rccl_child", how, kind], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    last = [l for l in r.stdout.splitlines() if l.startswith("OK ")]
    assert last, r.stdout[-2000:]
    _, ntrials, worst, nranks = last[-1].split()
    assert int(ntrials) >= 3 and float(worst) <= 1e-10 and int(nranks) == 1
    return r


@pytest.mark.parametrize("kind", ["sparse", "device", "dense"])
def test_set_communicator_with_rccl_at_world_size_one(gpu, kind):
    """dogleg_amd_set_communicator(0, 1, 0, id): partition / row shard installed, ncclCommInitRank, every sum of the
    solve an ncclAllReduce on the solve's stream; the trace is the oracle's"""
    _child("api", kind)


@pytest.mark.parametrize("kind", ["sparse", "device"])
def test_environment_contract_with_rccl_at_world_size_one(gpu, kind, tmp_path):
    """DOGLEG_AMD_WORLD_SIZE=1 + DOGLEG_AMD_FORCE_COMM: the path a re-linked program takes under a launcher -- rank 0
    removes what it finds at DOGLEG_AMD_RCCL_ID_FILE, writes this launch's id there, the communicator is made once
    per process and shared with the solve"""
    idf = tmp_path / "rccl.id"
    idf.write_bytes(b"\x5a" * 128)              # a file an earlier launch (of the old format) left behind
    _child("env", kind, {"DOGLEG_AMD_WORLD_SIZE": "1", "DOGLEG_AMD_FORCE_COMM": "1", "DOGLEG_AMD_RANK": "0",
                         "DOGLEG_AMD_LOCAL_RANK": "0", "DOGLEG_AMD_RCCL_ID_FILE": str(idf), "DOGLEG_AMD_RUN_ID": "t-%d" % os.getpid()})
    rec = idf.read_bytes()
    assert len(rec) == 144 and rec[128:136] == b"DLGAMD01" and rec[:128] != b"\x5a" * 128
    assert not os.path.exists(str(idf) + ".tmp")
