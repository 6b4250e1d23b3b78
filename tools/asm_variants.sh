#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "-DDLG_ASM_U=4" "-DDLG_ASM_U=2" "-DDLG_ASM_U=4 -DDLG_ASM_WPE=6" "-DDLG_ASM_U=2 -DDLG_ASM_WPE=6" "-DDLG_ASM_U=2 -DDLG_ASM_WPE=8" "-DDLG_ASM_U=8"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $V -Iinclude -c libdogleg_amd/csrc/kernels_sparse.hip -o libdogleg_amd/csrc/_obj/kernels_sparse.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libdogleg_amd/libdogleg_amd.so libdogleg_amd/csrc/_obj/*.o
  for rk in 32 16; do
  echo "$V RUN_KG=$rk"; DOGLEG_AMD_RUN_KG=$rk timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_ms_per_step']['K4_kernel'], d['phases_ms_per_step']['K4_total'])"
  done
done
