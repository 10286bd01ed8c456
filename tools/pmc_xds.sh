#!/bin/bash
# SQ counters of the streaming expand+depthwise kernel on one shape (one rocprofv3 --pmc pass, <= 8 SQ counters).
# usage (GPU box, repo root): bash tools/pmc_xds.sh <tag> "<B Cin Cexp rate parts>" [counters...]
tag=$1; args=$2; shift 2
ctrs=("$@")
[ ${#ctrs[@]} -eq 0 ] && ctrs=(SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcx_$tag
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d gpurun_out/pmcx_$tag -o p -- python3 tools/xds_one.py $args > gpurun_out/pmcx_$tag.log 2>&1
f=$(find gpurun_out/pmcx_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "xdw_" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    base = sum(d.get("SQ_WAVE_CYCLES", [1])) / max(len(d.get("SQ_WAVE_CYCLES", [1])), 1)
    for c, v in d.items():
        a = sum(v) / len(v)
        print("   %-24s avg %16.0f  %6.1f%% of wave cycles" % (c, a, 100 * a / base))
PY
rm -rf gpurun_out/pmcx_$tag
