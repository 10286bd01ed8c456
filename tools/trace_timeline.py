#!/usr/bin/env python3
"""Per-launch timeline from a rocprofv3 --kernel-trace CSV: usage trace_timeline.py <kernel_trace.csv> [n_last_launches | marker-kernel-substring].
Lists the launches of the LAST step (from the last occurrence of the step's first kernel) in start order with duration, the gap to the
previous kernel's end on any stream, and grid size; then totals: busy time (union of intervals), sum of durations, idle gaps."""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")),
                     r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "stem_mfma_kernel"
starts = [i for i, r in enumerate(rows) if marker in r[2]]
if not starts:
    sys.exit("marker kernel %r not found" % marker)
first = starts[-1]
end = len(rows)
step = rows[first:end]
t0 = step[0][0]
prev_end = t0
busy = 0
cover_end = t0
tot = 0
print("%9s %8s %7s %9s %6s  %s" % ("start_us", "dur_us", "gap_us", "grid", "queue", "kernel"))
for s, e, name, grid, wg, q in step:
    gap = (s - cover_end) / 1e3
    short = name.replace("ams::", "").replace("void ", "")
    short = short.split("(")[0][:70]
    print("%9.1f %8.1f %7.1f %9s %6s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, grid, q, short))
    tot += e - s
    if e > cover_end:
        busy += e - max(s, cover_end)
        cover_end = e
print("launches %d  span %.1f us  busy(union) %.1f us  sum %.1f us  idle %.1f us" % (len(step), (cover_end - t0) / 1e3, busy / 1e3, tot / 1e3,
                                                                                  (cover_end - t0 - busy) / 1e3))
