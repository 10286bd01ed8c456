#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -q > gpurun_out/r06_pytest_gpu.txt 2>&1
tail -3 gpurun_out/r06_pytest_gpu.txt
python3 __graft_entry__.py smoke 2>&1 | tail -2
python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -c 1500 gpurun_out/r06_bench.json
