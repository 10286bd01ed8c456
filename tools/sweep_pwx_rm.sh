#!/bin/bash
# row-group sweep of the split-bf16 GEMM (AMS_PWX_FORCE = "RM,NT") on the stride-16 project / head shapes: does a finer row tile
# (more, smaller blocks) beat the tail of 1.4 rounds of 128-row blocks?  32 frames = 68640 rows
M=${1:-68640}
for shape in "960 160" "960 320" "576 96" "576 160" "384 64" "384 96" "320 256" "256 256" "192 64"; do
  set -- $shape
  for f in default 1,5 1,6 1,4 1,3 1,2 4,5 4,4 4,3; do
    if [ "$f" = default ]; then unset AMS_PWX_FORCE; else export AMS_PWX_FORCE=$f; fi
    echo -n "force=$f  "
    python3 tools/bench_kernel.py $M $1 $2 split3 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
