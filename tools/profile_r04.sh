#!/bin/bash
# Round-4 artefacts on the GPU box (repo root): usage  bash tools/profile_r04.sh [part ...]   parts: bench infer train pmc sq (default: all)
# -> gpurun_out/r04_*  (copied to profiles/ by hand after a look)
parts=${@:-bench infer train pmc sq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stats() {  # <dir> <out.csv>
  s=$(find $1 -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $2
}
for part in $parts; do case $part in
bench)
  python3 bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err ;;
infer)
  # the ONE-stream plan (what `roofline` prices per kernel) and the plan the headline times (two parts on two streams at 32 frames)
  rm -rf gpurun_out/p4i
  AMS_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4i/one -o p -- python3 bench.py --only-timed --steps 20 --warmup 3 --settle 0 --windows 1 > gpurun_out/r04_prof_infer.log 2>&1
  stats gpurun_out/p4i/one gpurun_out/r04_infer_kernel_stats.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4i/dual -o p -- python3 bench.py --only-timed --steps 20 --warmup 3 --settle 0 --windows 1 > gpurun_out/r04_prof_infer_dual.log 2>&1
  stats gpurun_out/p4i/dual gpurun_out/r04_infer_kernel_stats_dual.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4i/b1 -o p -- python3 tools/infer_loop.py 1 512 200 20 > gpurun_out/r04_prof_infer_b1.log 2>&1
  stats gpurun_out/p4i/b1 gpurun_out/r04_infer_b1_kernel_stats.csv
  rm -rf gpurun_out/p4i ;;
train)
  rm -rf gpurun_out/p4t
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4t -o t -- python3 tools/train_loop.py 8 512 5 3 > gpurun_out/r04_prof_train.log 2>&1
  stats gpurun_out/p4t gpurun_out/r04_train_kernel_stats.csv
  f=$(find gpurun_out/p4t -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_timeline.py $f > gpurun_out/r04_train_timeline.txt 2>&1
  rm -rf gpurun_out/p4t ;;
pmc)
  # HBM traffic: two separate counter passes each (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
  bash tools/pmc_traffic.sh r04 > gpurun_out/r04_pmc_infer.log 2>&1           # -> gpurun_out/pmc_traffic_r04.json (32 frames, one-stream plan)
  rm -rf gpurun_out/pmc_r04t
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r04t/fetch -o p -- python3 tools/train_loop.py 8 512 2 1 > gpurun_out/r04_pmc_train_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r04t/write -o p -- python3 tools/train_loop.py 8 512 2 1 > gpurun_out/r04_pmc_train_write.log 2>&1
  alg=$(python3 -c "import json;print(json.loads(open('gpurun_out/r04_bench.json').read().strip().splitlines()[-1])['distill']['roofline']['alg_bytes_per_step'])" 2>/dev/null)
  python3 tools/pmc_train_traffic.py gpurun_out/pmc_r04t 3 gpurun_out/r04_train_pmc_traffic.json $alg
  rm -rf gpurun_out/pmc_r04t gpurun_out/pmc_r04 ;;
sq)
  # SQ counters of the round's new fine-tune kernels (unit-correct tables: tools/pmc_cmd.sh)
  bash tools/pmc_cmd.sh r04_xdw xdw_ python3 tools/train_loop.py 8 512 1 1
  bash tools/pmc_cmd.sh r04_dwbn dw3x3_ python3 tools/train_loop.py 8 512 1 1
  cat gpurun_out/sq_r04_xdw.txt gpurun_out/sq_r04_dwbn.txt > gpurun_out/r04_train_sq_counters.txt
  bash tools/pmc_cmd.sh r04_gemm "pw_gemm_bf16x3_l<2, 5, 1" python3 tools/infer_loop.py 32 512 2 1 0
  cp gpurun_out/sq_r04_gemm.txt gpurun_out/r04_split_gemm_sq_counters.txt ;;
esac; done
ls -la gpurun_out/r04_* | head -30
