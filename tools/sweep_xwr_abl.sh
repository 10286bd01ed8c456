#!/bin/bash
# MEASUREMENT ONLY: the weight-register expand + depthwise kernel (160 -> 960, fp16 parts) inside the 32-frame step with parts switched off
# (AMS_XWR_ABL bits: 1 no operand loads, 2 no MFMAs, 4 no depthwise arithmetic, 8 no result stores, 16 no ring stores).  Wrong results by design.
# needs the measurement build: make -C ams_amd/csrc measure (libams_hip_measure.so; the product library has no ablated kernels)
export AMS_HIP_LIB=${AMS_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ams_amd/libams_hip_measure.so}
[ -f "$AMS_HIP_LIB" ] || { echo "build it first: make -C ams_amd/csrc measure"; exit 1; }
for a in 0 1 2 3 4 8 12 16 19 28 31; do
  echo -n "abl=$a  "
  AMS_XWR_ABL=$a AMS_DUAL_STREAM=0 python3 bench.py --no-train --no-stream --no-api --no-cpu --no-parity --no-bf16 --dump-layers --steps 5 --windows 1 2>&1 | grep -v "^{" | grep xdw_wreg | head -1
done
