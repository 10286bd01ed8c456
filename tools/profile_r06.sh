#!/bin/bash
# Round-6 artefacts on the GPU box (repo root): usage  bash tools/profile_r06.sh [part ...]   parts: bench infer train pmc sq phases (default: all)
# -> gpurun_out/r06_*  (copied to profiles/ by hand after a look)
parts=${@:-bench infer train pmc sq phases}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stats() {  # <dir> <out.csv>
  s=$(find $1 -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $2
}
for part in $parts; do case $part in
bench)
  python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err ;;
infer)
  # the ONE-stream plan (what `roofline` prices per kernel) and the plan the headline times (two parts on two streams at 32 frames)
  rm -rf gpurun_out/p6i
  AMS_DUAL_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p6i/one -o p -- python3 bench.py --only-timed --steps 20 --warmup 3 --settle 0 --windows 1 > gpurun_out/r06_prof_infer.log 2>&1
  stats gpurun_out/p6i/one gpurun_out/r06_infer_kernel_stats.csv
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p6i/dual -o p -- python3 bench.py --only-timed --steps 20 --warmup 3 --settle 0 --windows 1 > gpurun_out/r06_prof_infer_dual.log 2>&1
  stats gpurun_out/p6i/dual gpurun_out/r06_infer_kernel_stats_dual.csv
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p6i/b1 -o p -- python3 tools/infer_loop.py 1 512 200 20 > gpurun_out/r06_prof_infer_b1.log 2>&1
  stats gpurun_out/p6i/b1 gpurun_out/r06_infer_b1_kernel_stats.csv
  rm -rf gpurun_out/p6i ;;
train)
  rm -rf gpurun_out/p6t
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p6t -o t -- python3 tools/train_loop.py 8 512 5 3 > gpurun_out/r06_prof_train.log 2>&1
  stats gpurun_out/p6t gpurun_out/r06_train_kernel_stats.csv
  f=$(find gpurun_out/p6t -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_timeline.py $f > gpurun_out/r06_train_timeline.txt 2>&1
  rm -rf gpurun_out/p6t ;;
pmc)
  # HBM traffic: two separate counter passes each (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
  bash tools/pmc_traffic.sh r06 > gpurun_out/r06_pmc_infer.log 2>&1           # -> gpurun_out/pmc_traffic_r06.json (32 frames, one-stream plan)
  rm -rf gpurun_out/pmc_r06t
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r06t/fetch -o p -- python3 tools/train_loop.py 8 512 2 1 > gpurun_out/r06_pmc_train_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r06t/write -o p -- python3 tools/train_loop.py 8 512 2 1 > gpurun_out/r06_pmc_train_write.log 2>&1
  alg=$(python3 -c "import json;print(json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])['distill']['roofline']['alg_bytes_per_step'])" 2>/dev/null)
  python3 tools/pmc_train_traffic.py gpurun_out/pmc_r06t 3 gpurun_out/r06_train_pmc_traffic.json $alg
  rm -rf gpurun_out/pmc_r06t gpurun_out/pmc_r06 ;;
sq)
  # SQ counters (unit-correct tables: tools/pmc_cmd.sh) of the round's inference kernels: the dominant streaming kernel, the fp16 GEMM in its
  # 12-wave full-width form, a whole-block kernel and the first block
  bash tools/pmc_cmd.sh r06_xwr "xdw_wreg_kernel" python3 tools/infer_loop.py 32 512 2 1 0
  bash tools/pmc_cmd.sh r06_gemm "pw_gemm_f16x3_l<1, 10, 1" python3 tools/infer_loop.py 32 512 2 1 0
  bash tools/pmc_cmd.sh r06_blk "block_kernel<2, 1, 2, 4, 8" python3 tools/infer_loop.py 32 512 2 1 0
  bash tools/pmc_cmd.sh r06_fb "first_block" python3 tools/infer_loop.py 32 512 2 1 0
  cat gpurun_out/sq_r06_xwr.txt gpurun_out/sq_r06_gemm.txt gpurun_out/sq_r06_blk.txt gpurun_out/sq_r06_fb.txt > gpurun_out/r06_infer_sq_counters.txt ;;
phases)
  # in-kernel phase clocks (s_memtime laps summed per wave; ams_debug_phase_cycles): the walking first block, the weight-register streaming kernel
  # inside the 32-frame step, the five whole-block shapes alone
  { echo "# shader-clock cycles per wave and phase (tools/fb_phases.py, tools/xwr_phases.py, tools/block_one.py with AMS_BLK_TIMED=1)"
    python3 tools/fb_phases.py 32
    python3 tools/xwr_phases.py 32
    for sh in "256 512 16 96 24 2 0" "128 256 24 144 24 1 1" "128 256 24 144 32 2 0" "64 128 32 192 32 1 1" "64 128 32 192 64 2 0"; do
      BLK_F16=1 AMS_BLK_TIMED=1 python3 tools/block_one.py 32 $sh
    done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_phase_clocks.txt ;;
esac; done
ls -la gpurun_out/r06_* | head -30
