#!/bin/bash
# tile sweep of the three-part split GEMM at the fine-tune step's row count (8 frames = 17160 rows at the output stride): AMS_PWX_FORCE = "RM,NT"
M=${1:-17160}
for shape in "960 160" "576 96" "384 64" "160 960" "96 576" "64 384" "960 320" "320 256"; do
  set -- $shape
  for f in default 1,2 1,3 1,4 1,5 1,6 2,2 2,3 2,4 2,5 2,6 2,8 2,10 4,3 4,4 4,5; do
    if [ "$f" = default ]; then unset AMS_PWX_FORCE; else export AMS_PWX_FORCE=$f; fi
    echo -n "force=$f  "
    python3 tools/bench_kernel.py $M $1 $2 split3 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
