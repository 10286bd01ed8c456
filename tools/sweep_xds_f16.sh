#!/bin/bash
# The LDS-weight streaming kernel (64 / 96-channel stride-16 blocks) under the fp16 forms: chunk width and blocks per chunk
# (AMS_XDS_FORCE = "tiles,row segments,column strips,E-waves,D-waves,blocks per chunk"); kernel times from rocprofv3 of the 32-frame one-stream loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in ${@:-none 4,0,0,4,4,0 2,0,0,4,4,1024 2,0,0,4,4,4096 4,0,0,4,4,1024 2,2,0,4,4,0 4,2,0,4,4,0}; do
  if [ $f = none ]; then unset AMS_XDS_FORCE; else export AMS_XDS_FORCE=$f; fi
  rm -rf gpurun_out/xdsf
  timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/xdsf -o p --output-format csv -- python3 tools/infer_loop.py 32 512 6 2 0 > /dev/null 2>&1
  python3 - $f <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/xdsf/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if 'xdw_stream' in r['Name']:
        print('force=%-16s %-66s %3s x %7.1f us' % (sys.argv[1], r['Name'][10:76], r['Calls'], float(r['AverageNs'])/1e3)); tot+=float(r['TotalDurationNs'])
print('force=%-16s total %.1f us per step' % (sys.argv[1], tot/1e3/8))
PY
done
rm -rf gpurun_out/xdsf
