#!/bin/bash
# SQ counters of the dense JtJ kernel k_syrk_lower<64> (config #2; rocprofv3 --pmc, counters only beside --kernel-trace)
export TMPDIR=/tmp; cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${1:-sqd}; mkdir -p $out
p=1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$p -o p -- python3 tools/dense_k4.py > $out/pass$p.log 2>&1
  python3 tools/pmc_kernel.py $out/pass$p k_syrk_lower | tee -a $out/sq_syrk.txt
  p=$((p+1))
done
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete; find $out -name "*counter_collection.csv" -delete
