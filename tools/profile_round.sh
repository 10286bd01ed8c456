#!/bin/bash
# Round artefacts on the GPU box (repo root): full bench JSON, rocprofv3 kernel-trace summary of the timed inference loop
# and of one fine-tune leg, PMC traffic passes.  usage: bash tools/profile_round.sh <tag>   -> gpurun_out/<tag>_*
tag=${1:-rXX}
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
# kernel statistics of the ONE-stream plan: the launches the bench line prices per kernel (`roofline`); the headline runs the same kernels as
# two half-batches on two streams where that is faster (AMS_OPT_DUAL_STREAM), in half-size launches that overlap
AMS_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag/infer -o p -- python3 bench.py --only-timed --steps 20 --warmup 3 --settle 0 > gpurun_out/${tag}_prof_infer.log 2>&1
db=$(find gpurun_out/prof_$tag/infer -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" gpurun_out/${tag}_infer_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag/train -o p -- python3 tools/profile_train.py 8 512 > gpurun_out/${tag}_prof_train.log 2>&1
db=$(find gpurun_out/prof_$tag/train -name "*.db" | head -1)
python3 tools/rocpd_stats.py "$db" gpurun_out/${tag}_train_kernel_stats.csv
bash tools/pmc_traffic.sh $tag > gpurun_out/${tag}_pmc.log 2>&1      # -> gpurun_out/pmc_traffic_$tag.json (copy to profiles/<round>_b<batch>_pmc_traffic.json)
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_$tag
head -5 gpurun_out/${tag}_infer_kernel_stats.csv
