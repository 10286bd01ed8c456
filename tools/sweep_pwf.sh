#!/bin/bash
# tile sweep of the two-fp16-part GEMM (AMS_PWX_FORCE = "RM,NT") on the stride-16 project / head shapes; 32 frames = 68640 rows
# usage: tools/sweep_pwf.sh [M] [f16|f16p]
M=${1:-68640}
MODE=${2:-f16p}
for shape in "960 160" "960 320" "576 96" "576 160" "384 64" "384 96" "320 256" "256 256"; do
  set -- $shape
  for f in default 2,10 2,8 2,6 2,5 2,4 2,3 4,4 4,3 1,5 1,4; do
    if [ "$f" = default ]; then unset AMS_PWX_FORCE; else export AMS_PWX_FORCE=$f; fi
    echo -n "force=$f  "
    python3 tools/bench_kernel.py $M $1 $2 $MODE 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
