#!/bin/bash
# A/B of the first block's forms on the 32-frame and the one-frame step + the kernels' rocprofv3 times:
# AMS_FB_WALK=0 one tile per block; -2 interior tiles walking, border tiles one per block; default: both walking
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in 0 -2 -1; do
  if [ $w = -1 ]; then unset AMS_FB_WALK; else export AMS_FB_WALK=$w; fi
  echo "== AMS_FB_WALK=$w"
  python tools/infer_loop.py 32 512 40 10
  python tools/infer_loop.py 1 512 200 20
  rm -rf gpurun_out/fbw; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/fbw -o p --output-format csv -- python3 tools/infer_loop.py 32 512 20 5 0 > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/fbw/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'first_block' in r['Name']: print('   %-70s %4s launches %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
rm -rf gpurun_out/fbw
