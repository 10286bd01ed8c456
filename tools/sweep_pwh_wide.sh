#!/bin/bash
# EXPERIMENT: full-width (160-column) tiles of the fp16 GEMM in 8- / 12-wave blocks against the 80-wide default; pre-packed operand
M=${1:-68640}
for shape in "960 160" "576 160" "960 320"; do
  set -- $shape
  for cfg in "4,2 2,5" "12,2 2,5" "12,2 1,10" "8,2 1,10" "8,2 2,10" "12,3 1,10" "8,3 2,10"; do
    set -- $shape
    v=${cfg% *}; f=${cfg#* }
    export AMS_PWH_VARIANT=$v AMS_PWX_FORCE=$f
    echo -n "variant=$v tile=$f  "
    python3 tools/bench_kernel.py $M $1 $2 f16p 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
