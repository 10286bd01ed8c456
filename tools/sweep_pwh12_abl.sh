#!/bin/bash
# MEASUREMENT ONLY (libams_hip_measure.so): the 12-wave full-width fp16 GEMM (68640 x 960 -> 160, fp16 pairs in) with parts of its stage loop switched off
# (AMS_PWH_ABL bits: 1 no activation loads, 2 no weight loads / LDS stores, 8 no MFMAs, 16 the activation bytes requested in FULL-LINE lane order: a request
# instruction covers 8 rows x 128 bytes instead of 16 rows x 64 — what the texture path would see behind a lane-order hop).  Wrong results by design.
cd "$(dirname "$0")/.."
export AMS_HIP_LIB=${AMS_HIP_LIB:-$PWD/ams_amd/libams_hip_measure.so}
[ -f "$AMS_HIP_LIB" ] || { echo "build it first: make -C ams_amd/csrc measure"; exit 1; }
export AMS_PWH_VARIANT=12,3      # (the 160-column layers run 10-wave blocks since round 6: the ablated forms are 12-wave)
for a in ${ABLS:-0 16 1 2 3 8 9 10 11 18 24 26 32 35 48}; do
  export AMS_PWH_ABL=$a
  echo -n "abl=$a  "
  python3 tools/bench_kernel.py ${1:-68640} ${2:-960} ${3:-160} f16p 2>&1 | grep -v amdgpu.ids | tail -1
done
