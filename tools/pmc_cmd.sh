#!/bin/bash
# SQ counters of the kernels of ANY command whose name contains <pattern>, three rocprofv3 --pmc passes of <= 8 counters each.
# Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count QUAD-cycles summed over waves (share = x / SQ_WAVE_CYCLES);
# SQ_BUSY_CYCLES and SQ_VALU_MFMA_BUSY_CYCLES count CYCLES: SQ_BUSY_CYCLES summed over the 32 SQ instances of the chip (one per shader engine:
# SQ_BUSY_CYCLES / 32 ~ the kernel's duration in cycles), SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs (= MFMAs x 16 cycles for the
# 16x16x32 bf16 form).  The MFMA pipe's busy share is therefore MFMA_BUSY / (1024 x SQ_BUSY / 32) = MFMA_BUSY / (32 x SQ_BUSY);
# SQ_INSTS_* count instructions.
# usage (GPU box, repo root): bash tools/pmc_cmd.sh <tag> <kernel substring> <program> [args...]    (program directly: no env / bash -c hop)
tag=$1; pat=$2; shift 2
groups=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS"
        "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"
        "SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sq_$tag.txt
: > $out
g=0
for ctrs in "${groups[@]}"; do
  g=$((g+1))
  rm -rf gpurun_out/pmcc_$tag
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmcc_$tag -o p -- "$@" > gpurun_out/pmcc_$tag.log 2>&1
  f=$(find gpurun_out/pmcc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$pat" $g >> $out <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
QUAD = ("SQ_WAVE_CYCLES", "SQ_WAIT_", "SQ_ACTIVE_INST_", "SQ_INST_CYCLES_")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    n = len(d.get("SQ_WAVE_CYCLES", [1]))
    wave = sum(d.get("SQ_WAVE_CYCLES", [1])) / n
    busy = sum(d.get("SQ_BUSY_CYCLES", [0])) / max(len(d.get("SQ_BUSY_CYCLES", [1])), 1)
    print("%s   [pass %s, %d launches]" % (k, sys.argv[3], n))
    for c, v in d.items():
        a = sum(v) / len(v)
        if c.startswith(QUAD):
            print("   %-26s %16.0f quad-cycles  %6.1f %% of wave quad-cycles" % (c, a, 100 * a / wave))
        elif c in ("SQ_BUSY_CYCLES",):
            print("   %-26s %16.0f cycles (summed over SQs)" % (c, a))
        elif c == "SQ_VALU_MFMA_BUSY_CYCLES":
            print("   %-26s %16.0f cycles (summed over SIMDs)  pipe busy %5.1f %% = x / (32 x SQ_BUSY_CYCLES)" % (c, a, 100 * a / (32 * busy) if busy else 0.0))
        else:
            print("   %-26s %16.0f" % (c, a))
PY
done
rm -rf gpurun_out/pmcc_$tag
cat $out
