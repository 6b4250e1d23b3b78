#!/bin/bash
# tools only: A/B a compile-time variant of ONE source file of the library.
#   tools/variant_lib.sh build sparse_assemble.hip -DDLG_ASM_PREFETCH     (in the container) -> tools/micro/libvar.so
#   tools/variant_lib.sh run [bench args]                                  (on the GPU box): bench.py with that library
cd "$(dirname "$0")/.." || exit 1
if [ "$1" = build ]; then
  src=$2; shift 2
  python3 -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
  extra=""; case $src in sparse_assemble.hip|sparse_factor.hip|dense_diag.hip|sparse_leaf.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $extra "$@" -Iinclude -c libdogleg_amd/csrc/$src -o /tmp/var_$src.o || exit 1
  objs=$(ls libdogleg_amd/csrc/_obj/*.o | grep -v "/$src.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/micro/libvar.so $objs /tmp/var_$src.o
  exit $?
fi
shift
DLG_LIB=tools/micro/libvar.so python3 - "$@" <<'PY'
import os, sys
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
from libdogleg_amd import capi
capi.LIB_PATH = os.path.abspath(os.environ["DLG_LIB"])
import bench
bench.main()
PY
