#!/bin/bash
# usage: tools/pmc_run2.sh <outdir-under-gpurun_out> <kernel-pattern> ; SQ counter groups, separate passes
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$1; pat=$2
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/g$i -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out.g$i.log 2>&1
  python tools/pmc_kernel.py $out/g$i $pat || tail -3 $out.g$i.log
done
