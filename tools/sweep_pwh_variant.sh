#!/bin/bash
# EXPERIMENT: the fp16 GEMM with wider blocks (waves per block) and deeper operand rings, AMS_PWH_VARIANT=<waves>,<depth>; pre-packed operand
M=${1:-68640}
for shape in "960 160" "960 320" "576 96" "576 160" "384 64"; do
  set -- $shape
  for v in 4,2 4,3 4,4 8,2 8,3 12,2 12,3; do
    for f in 2,5 2,4; do
      export AMS_PWH_VARIANT=$v AMS_PWX_FORCE=$f
      echo -n "variant=$v tile=$f  "
      python3 tools/bench_kernel.py $M $1 $2 f16p 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
