#!/bin/bash
# the walking first block's phases in shader-clock cycles, as built (32) and under the ablations of tools/sweep_fb_abl.sh
for a in ${@:-32 33 36 34 40 48 63}; do AMS_FB_ABL=$a python tools/fb_phases.py 32 2>&1 | grep -v amdgpu.ids; done
