#!/bin/bash
# A/B of the first block's two forms (AMS_FB_WALK=0: one tile per block) on the 32-frame and the one-frame step + the kernel's rocprofv3 time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in 0 -1; do
  if [ $w = -1 ]; then unset AMS_FB_WALK; else export AMS_FB_WALK=$w; fi
  echo "== AMS_FB_WALK=$w"
  python tools/infer_loop.py 32 512 40 10
  python tools/infer_loop.py 32 512 40 10 0
  python tools/infer_loop.py 1 512 200 20
  python tools/infer_loop.py 2 512 200 20
done
unset AMS_FB_WALK
rm -rf gpurun_out/fbw; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/fbw -o p --output-format csv -- python3 tools/infer_loop.py 32 512 20 5 0 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/fbw/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
