#!/bin/bash
# SQ counters (three groups of <= 8) for the kernels of the timed inference loop whose name contains <pattern>.
# usage (GPU box, repo root): bash tools/pmc_groups.sh <tag> <kernel substring>
tag=$1; pat=$2
groups=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS"
        "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"
        "SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM")
for ctrs in "${groups[@]}"; do
  bash tools/pmc_kernel.sh $tag "$pat" $ctrs
done
