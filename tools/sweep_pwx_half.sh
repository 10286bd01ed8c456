#!/bin/bash
# tile sweep of the split-bf16 GEMM (AMS_PWX_FORCE = "RM,NT") at the half-batch row count of the two-stream plan (16 frames = 34320 rows)
M=34320
for shape in "960 160" "960 320" "576 96" "384 64" "320 256" "256 256"; do
  set -- $shape
  for f in default 2,4 2,5 2,6 2,10 1,4 1,5 1,6 4,4 4,5; do
    if [ "$f" = default ]; then unset AMS_PWX_FORCE; else export AMS_PWX_FORCE=$f; fi
    out=$(python3 tools/bench_kernel.py $M $1 $2 split3 2>&1 | grep -v amdgpu.ids | tail -1)
    echo "force=$f  $out"
  done
done
