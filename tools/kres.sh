#!/bin/bash
# usage: tools/kres.sh <file.hip> [grep pattern]: per-kernel registers / scratch / occupancy from hipcc's resource-usage remarks
cd "$(dirname "$0")/../ams_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
name=None; rec={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: name=m.group(1); rec[name]={}
    for key,short in (('VGPRs:','v'),('AGPRs','a'),('ScratchSize','scr'),('Occupancy','occ'),('LDS Size','lds')):
        if key in line and name: rec[name][short]=line.split(':')[-1].strip()
names=list(rec)
d=subprocess.run(['c++filt']+names,capture_output=True,text=True).stdout.strip().split('\n')
for n,dn in zip(names,d):
    dn=dn.split('(')[0].replace('void ams::','')
    print('%-60s'%dn, rec[n])
" | grep -E "${2:-.}"
