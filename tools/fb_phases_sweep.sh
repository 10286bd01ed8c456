#!/bin/bash
# the walking first block's phases in shader-clock cycles, as built (32) and under the ablations of tools/sweep_fb_abl.sh
# needs the measurement build: make -C ams_amd/csrc measure (libams_hip_measure.so; the product library has no ablated kernels)
export AMS_HIP_LIB=${AMS_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ams_amd/libams_hip_measure.so}
[ -f "$AMS_HIP_LIB" ] || { echo "build it first: make -C ams_amd/csrc measure"; exit 1; }
for a in ${@:-32 33 36 34 40 48 63}; do AMS_FB_ABL=$a python tools/fb_phases.py 32 2>&1 | grep -v amdgpu.ids; done
