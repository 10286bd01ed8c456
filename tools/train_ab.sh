#!/bin/bash
# A/B of the fine-tune step under option settings on ONE box: tools/train_ab.sh "VAR=val ..." "VAR=val ..." ...   (8 x 512x1024, 30 steps each, twice)
for rep in 1 2; do
  for cfg in "$@"; do
    echo -n "[$cfg] "
    env $cfg python3 tools/train_loop.py 8 512 30 5 2>&1 | grep "train step" | cut -c1-60
  done
done
