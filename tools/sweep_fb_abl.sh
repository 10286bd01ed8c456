#!/bin/bash
# What the walking first block's time is made of: AMS_FB_ABL=<bits> (wrong results): 1 no result stores, 2 no depthwise arithmetic, 4 no byte loads /
# table look-ups, 8 no project MFMAs, 16 no stem MFMAs.  Kernel time from rocprofv3 --kernel-trace --stats of a 32-frame one-stream loop.
# needs the measurement build: make -C ams_amd/csrc measure (libams_hip_measure.so; the product library has no ablated kernels)
export AMS_HIP_LIB=${AMS_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/ams_amd/libams_hip_measure.so}
[ -f "$AMS_HIP_LIB" ] || { echo "build it first: make -C ams_amd/csrc measure"; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for abl in ${@:-0 1 2 4 8 16 6 14 30 31}; do
  export AMS_FB_ABL=$abl
  rm -rf gpurun_out/fba
  timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/fba -o p --output-format csv -- python3 tools/infer_loop.py 32 512 6 2 0 > /dev/null 2>&1
  python3 - $abl <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/fba/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'first_block' in r['Name']: print('abl=%-3s %-50s %8.1f us' % (sys.argv[1], r['Name'][:50], float(r['AverageNs'])/1e3))
PY
done
rm -rf gpurun_out/fba
