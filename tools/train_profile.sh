#!/bin/bash
# rocprofv3 kernel trace of the fine-tune loop -> per-launch timeline of the last step + per-kernel stats
# usage (GPU box, repo root): bash tools/train_profile.sh <tag> [B] [H]
tag=$1; B=${2:-8}; H=${3:-512}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tt_$tag -o t -- python3 tools/train_loop.py $B $H 3 2 > gpurun_out/tt_$tag.log 2>&1
f=$(find gpurun_out/tt_$tag -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f > gpurun_out/timeline_$tag.txt 2>&1
s=$(find gpurun_out/tt_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp $s gpurun_out/kstats_$tag.csv
rm -rf gpurun_out/tt_$tag
tail -1 gpurun_out/tt_$tag.log; tail -1 gpurun_out/timeline_$tag.txt
