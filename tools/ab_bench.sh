#!/bin/bash
# A/B of two builds of the library on ONE box (run-to-run spread between boxes is +-2 %): alternates them n times.
# usage (GPU box, repo root): bash tools/ab_bench.sh <libA.so> <libB.so> [n] [bench args...]
A=$1; B=$2; n=${3:-3}; shift 3
for i in $(seq 1 $n); do
  for lib in $A $B; do
    v=$(AMS_HIP_LIB=$lib python3 bench.py --no-cpu --no-train --no-stream --no-bf16 --no-parity --no-profile "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$lib $v"
  done
done
