#!/bin/bash
# Any counters on the kernels of any command whose name contains <pattern>: one rocprofv3 --pmc pass per quoted counter group.
# usage (GPU box, repo root): bash tools/pmc_any.sh <tag> <kernel substring> "<group 1>" ["<group 2>" ...] -- <program> [args...]
tag=$1; pat=$2; shift 2
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag.txt
: > $out
for ctrs in "${groups[@]}"; do
  rm -rf gpurun_out/pmca_$tag
  timeout 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmca_$tag -o p -- "$@" > gpurun_out/pmca_$tag.log 2>&1
  f=$(find gpurun_out/pmca_$tag -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "no counter file for: $ctrs" >> $out; tail -5 gpurun_out/pmca_$tag.log >> $out; continue; fi
  python3 - "$f" "$pat" >> $out <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-36s %18.0f   (avg of %d launches)" % (c, sum(v) / len(v), len(v)))
PY
done
rm -rf gpurun_out/pmca_$tag
cat $out
