#!/bin/bash
# tile sweep of the split-bf16 GEMM on the stride-16 project / head shapes (AMS_PWX_FORCE = "RM,NT"); 32 frames = 68640 rows
M=${1:-68640}
for shape in "960 160" "960 320" "576 96" "576 160" "384 96" "320 256" "256 256" "160 960"; do
  set -- $shape
  for f in default 2,10 2,8 2,6 2,5 2,4 1,10; do
    if [ "$f" = default ]; then unset AMS_PWX_FORCE; else export AMS_PWX_FORCE=$f; fi
    echo -n "force=$f  "
    python3 tools/bench_kernel.py $M $1 $2 split3 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
