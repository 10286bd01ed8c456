#!/bin/bash
# Row segments per sub-image of the LDS-weight streaming kernels (AMS_XDS_FORCE = "tiles,row segments,column strips,E-waves,D-waves,blocks per chunk") at 32 and
# 16 frames (the two-stream plan launches 16): kernel times from the engine's per-launch profile of a one-stream step
cd "$GRAFT_REPO_ROOT"
for b in 32 16; do
for f in none 2,1,0,4,4,0 2,2,0,4,4,0 2,3,0,4,4,0 2,4,0,4,4,0 2,6,0,4,4,0 2,8,0,4,4,0; do
  if [ $f = none ]; then unset AMS_XDS_FORCE; else export AMS_XDS_FORCE=$f; fi
  echo -n "B=$b force=$f  "
  AMS_DUAL_STREAM=0 python3 bench.py --batch $b --no-train --no-stream --no-api --no-cpu --no-parity --no-bf16 --dump-layers --steps 10 --windows 1 2>&1 | grep -v "^{" | grep "xdw_stream" | python3 -c "
import sys,re
a={}
for l in sys.stdin:
    m=re.search(r'xdw_stream_kernel<(\d)', l); t=re.search(r'([0-9.]+) us', l)
    if m and t: a.setdefault(m.group(1), []).append(float(t.group(1)))
print('  '.join('KS=%s: %d x %.1f us' % (k, len(v), sum(v)/len(v)) for k,v in sorted(a.items())), ' total %.1f' % sum(sum(v) for v in a.values()))"
done
done
