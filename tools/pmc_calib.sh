#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_calib.sh   -> gpurun_out/pmc_calib.json + table on stdout
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_calib
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_calib/fetch -o p -- python3 tools/pmc_calib.py > gpurun_out/pmc_calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_calib/write -o p -- python3 tools/pmc_calib.py > gpurun_out/pmc_calib_write.log 2>&1
grep -E "read .* KB" gpurun_out/pmc_calib_fetch.log
python3 tools/pmc_traffic.py gpurun_out/pmc_calib gpurun_out/pmc_calib.json
