#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...> -- <python args for tools/bench_kernel.py>
# one rocprofv3 --pmc pass over the pointwise micro-benchmark; prints per-kernel averages of each counter
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_$tag
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o p -- python3 tools/bench_kernel.py "$@" > gpurun_out/pmc_$tag.log 2>&1
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "pw_gemm" not in k and "dw3x3" not in k: continue
    print(k)
    for c, v in d.items():
        print("   %-34s avg %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
