#!/bin/bash
# HBM traffic of every kernel of the bench step from the L2 memory-side counters, two separate rocprofv3 --pmc passes
# (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage (on the GPU box, repo root):  bash tools/pmc_traffic.sh <tag> [bench.py args]
# writes gpurun_out/pmc_<tag>/{fetch,write}/...counter_collection.csv and gpurun_out/pmc_traffic_<tag>.json
tag=$1; shift
args=("$@")
export AMS_DUAL_STREAM=0      # per-kernel traffic of the one-stream plan, whose launches bench.py prices in `roofline`
[ ${#args[@]} -eq 0 ] && args=(--steps 3 --warmup 1 --settle 0 --windows 1 --only-timed)      # default batch of bench.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_$tag
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$tag/fetch -o p -- python3 bench.py "${args[@]}" > gpurun_out/pmc_${tag}_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$tag/write -o p -- python3 bench.py "${args[@]}" > gpurun_out/pmc_${tag}_write.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_$tag gpurun_out/pmc_traffic_$tag.json
