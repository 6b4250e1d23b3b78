#!/bin/bash
# tools only: per-phase clocks of workgroup 0 of every k_factor_level launch.
# Build (in the container):  tools/prof_factor.sh build   -> gpurun_out is not used; tools/micro/libprof.so
# Run (on the GPU box):      tools/prof_factor.sh run [bench args]
cd "$(dirname "$0")/.." || exit 1
if [ "$1" = build ]; then
  python3 -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DDLG_FL_PROFILE \
    -Iinclude -c libdogleg_amd/csrc/sparse_factor.hip -o /tmp/sparse_factor_prof.o 2>/tmp/prof_build.err || exit 1
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDLG_FL_PROFILE \
    -Iinclude -c libdogleg_amd/csrc/sparse_solve.hip -o /tmp/sparse_solve_prof.o 2>/dev/null || exit 1
  objs=$(ls libdogleg_amd/csrc/_obj/*.o | grep -v sparse_factor.hip.o | grep -v sparse_solve.hip.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/micro/libprof.so $objs /tmp/sparse_factor_prof.o /tmp/sparse_solve_prof.o
  exit $?
fi
shift
DLG_LIB=tools/micro/libprof.so python3 - "$@" <<'PY'
import os, sys, ctypes
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"] + sys.argv[1:]
from libdogleg_amd import capi
capi.LIB_PATH = os.path.abspath(os.environ["DLG_LIB"])
import bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
capi.lib().dlg_fl_profile_dump(int(os.environ.get("DLG_FL_DUMP_N", "16")))
capi.lib().dlg_bw_profile_dump(16)
PY
