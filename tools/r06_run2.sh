#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_soft_teacher.py tests/test_gpu_multi.py -x -q 2>&1 | tail -15
echo "== role clocks, taps-ahead"
python3 tools/xwr_phases.py 32 2>&1 | grep -v amdgpu.ids | tail -4
echo "== role clocks, no taps-ahead"
AMS_XWR_NO_PRE=1 python3 tools/xwr_phases.py 32 2>&1 | grep -v amdgpu.ids | tail -4
