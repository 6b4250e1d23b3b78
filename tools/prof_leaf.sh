#!/bin/bash
# tools only: phase clocks of the leaf-front workgroups (sparse_leaf.hip, -DDLG_LF_PROFILE).
# Build (in the container):  tools/prof_leaf.sh build   -> tools/micro/liblfprof.so
# Run (on the GPU box):      tools/prof_leaf.sh run [bench args]
cd "$(dirname "$0")/.." || exit 1
if [ "$1" = build ]; then
  python3 -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DDLG_LF_PROFILE \
    -Iinclude -c libdogleg_amd/csrc/sparse_leaf.hip -o /tmp/sparse_leaf_prof.o 2>/tmp/lfprof_build.err || { cat /tmp/lfprof_build.err; exit 1; }
  objs=$(ls libdogleg_amd/csrc/_obj/*.o | grep -v sparse_leaf.hip.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/micro/liblfprof.so $objs /tmp/sparse_leaf_prof.o
  exit $?
fi
shift
DLG_LIB=tools/micro/liblfprof.so python3 - "$@" <<'PY'
import os, sys, ctypes
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"] + sys.argv[1:]
from libdogleg_amd import capi
capi.LIB_PATH = os.path.abspath(os.environ["DLG_LIB"])
import bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
capi.lib().dlg_lf_profile_dump(4096)
PY
