#!/bin/bash
# MEASUREMENT ONLY: the fp16 GEMM (68640 x 960 -> 160, pre-packed operand, 128 x 80 tiles) with parts of its stage loop switched off
# (AMS_PWH_ABL bits: 1 no activation loads, 2 no weight loads / LDS stores, 4 no barrier, 8 no MFMAs).  The results are wrong by design.
# needs the measurement build: make -C ams_amd/csrc measure (libams_hip_measure.so; the product library has no ablated kernels)
export AMS_HIP_LIB=${AMS_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ams_amd/libams_hip_measure.so}
[ -f "$AMS_HIP_LIB" ] || { echo "build it first: make -C ams_amd/csrc measure"; exit 1; }
M=${1:-68640}; K=${2:-960}; N=${3:-160}
export AMS_PWX_FORCE=2,5
for a in 0 1 2 3 4 6 7 8 9 10 11 15; do
  export AMS_PWH_ABL=$a
  echo -n "abl=$a  "
  python3 tools/bench_kernel.py $M $K $N f16p 2>&1 | grep -v amdgpu.ids | tail -1
done
