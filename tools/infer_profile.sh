#!/bin/bash
# rocprofv3 kernel stats of the frozen-inference loop: usage (GPU box, repo root): bash tools/infer_profile.sh <tag> [B] [dual]
tag=$1; B=${2:-32}; dual=${3:-0}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ip_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ip_$tag -o t -- python3 tools/infer_loop.py $B 512 10 3 $dual > gpurun_out/ip_$tag.log 2>&1
s=$(find gpurun_out/ip_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp $s gpurun_out/istats_$tag.csv
rm -rf gpurun_out/ip_$tag
tail -1 gpurun_out/ip_$tag.log
