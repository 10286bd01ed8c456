#!/bin/bash
# SQ counters per kernel of the timed inference loop (one rocprofv3 --pmc pass, <= 8 SQ counters).
# usage (GPU box, repo root): bash tools/pmc_kernel.sh <tag> <kernel substring> [counters...]
tag=$1; pat=$2; shift 2
ctrs=("$@")
[ ${#ctrs[@]} -eq 0 ] && ctrs=(SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmck_$tag
export AMS_DUAL_STREAM=0      # per-kernel counters of the one-stream plan (the two-stream plan runs the same kernels in half-size launches)
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d gpurun_out/pmck_$tag -o p -- python3 bench.py --steps 2 --warmup 1 --settle 0 --only-timed > gpurun_out/pmck_$tag.log 2>&1
f=$(find gpurun_out/pmck_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    base = sum(d.get("SQ_WAVE_CYCLES", [1])) / max(len(d.get("SQ_WAVE_CYCLES", [1])), 1)
    for c, v in d.items():
        a = sum(v) / len(v)
        print("   %-24s avg %16.0f  %6.1f%% of wave cycles" % (c, a, 100 * a / base))
PY
rm -rf gpurun_out/pmck_$tag
