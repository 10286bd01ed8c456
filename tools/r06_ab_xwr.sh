#!/bin/bash
# round 6: the weight-register kernel with / without its taps-ahead D-waves (AMS_XWR_NO_PRE), tests first
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_kernels.py -x -q -k "stream_f16 or stream or wreg" 2>&1 | tail -3
python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "small_batches or bench_batch or matches_oracle" 2>&1 | tail -3
for i in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export AMS_XWR_NO_PRE=1; else unset AMS_XWR_NO_PRE; fi
  echo "== NO_PRE=$v"
  python3 bench.py --no-train --no-stream --no-api --no-cpu --no-parity --no-bf16 --dump-layers --steps 20 --windows 3 2> gpurun_out/layers_$v.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  grep -i "xdw_wreg" gpurun_out/layers_$v.txt | head -3
done
done
