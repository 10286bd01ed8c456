#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
for v in 0 1 2; do
  export AMS_XWR_PRIO=$v
  echo -n "PRIO=$v  "
  AMS_DUAL_STREAM=0 python3 bench.py --no-train --no-stream --no-api --no-cpu --no-parity --no-bf16 --dump-layers --steps 10 --windows 1 2>&1 | grep -v "^{" | grep xdw_wreg | head -3 | grep -o "[0-9.]* us" | tr "\n" " "; echo
done
done
