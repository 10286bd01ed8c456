#!/bin/bash
# Round-2 artefacts (GPU box, repo root): tools/profile_round.sh + SQ counter passes for the kernels the round worked on.
bash tools/profile_round.sh r02
{
  echo "== SQ counters, timed inference loop (32 frames, 512x1024), one rocprofv3 --pmc pass per group; SQ_WAVE_CYCLES / WAIT / ACTIVE count quad-cycles"
  for pat in block_kernel first_block_kernel; do
    echo "---- $pat: waits / issue"
    bash tools/pmc_kernel.sh r02a $pat SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS
    echo "---- $pat: pipes"
    bash tools/pmc_kernel.sh r02b $pat SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES
  done
} > gpurun_out/r02_block_kernels_sq_counters.txt 2>&1
tail -5 gpurun_out/r02_bench.json | cut -c1-300
