#!/bin/bash
# SQ counters of the whole-block kernel on one shape (one rocprofv3 --pmc pass per counter group, <= 8 SQ counters each).
# usage (GPU box, repo root): bash tools/pmc_block.sh <tag> "<B H W Cin Cexp Cout stride res>"
tag=$1; args=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
groups=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS"
        "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"
        "SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM")
g=0
for ctrs in "${groups[@]}"; do
  rm -rf gpurun_out/pmcb_$tag
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmcb_$tag -o p -- python3 tools/block_one.py $args 4 > gpurun_out/pmcb_$tag.$g.log 2>&1
  f=$(find gpurun_out/pmcb_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
if not sys.argv[1]:
    print("no counter file (group rejected?)"); sys.exit(0)
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "block_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    base = sum(d.get("SQ_WAVE_CYCLES", [1])) / max(len(d.get("SQ_WAVE_CYCLES", [1])), 1)
    for c, v in d.items():
        a = sum(v) / len(v)
        print("   %-26s avg %16.0f  %7.2f%% of wave cycles" % (c, a, 100 * a / base))
PY
  g=$((g+1))
done
rm -rf gpurun_out/pmcb_$tag
