"""Aggregate the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into per-kernel HBM bytes per launch.

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B?  No: rocprofv3 reports them in KB (1 KB = 1024 B,
derived as TCC_EA0_RDREQ_32B*32 + (RDREQ - RDREQ_32B)*64 over 1024).  On gfx950 a 128-byte request is tallied as 64 B, so
wide coalesced streams read exactly half their bytes (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE is doubled here.  The
same under-count is assumed for WRITE_SIZE only after calibrating it on kernels of known output size (the `calib`
block of the JSON shows raw counter vs known bytes for those kernels).
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(dirname, counter):
    files = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)
    if not files:
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    root, out = sys.argv[1], sys.argv[2]
    f = per_kernel(root + "/fetch", "FETCH_SIZE")
    w = per_kernel(root + "/write", "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, []), w.get(k, [])
        res[k] = {
            "launches": max(len(fv), len(wv)),
            "fetch_kb_raw_avg": sum(fv) / len(fv) if fv else None,
            "write_kb_raw_avg": sum(wv) / len(wv) if wv else None,
        }
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["fetch_kb_raw_avg"] or 0) * kv[1]["launches"])[:40]:
        print("%-70s n=%4d fetch_raw %12.1f KB  write_raw %12.1f KB" % (k[:70], v["launches"], v["fetch_kb_raw_avg"] or -1, v["write_kb_raw_avg"] or -1))


if __name__ == "__main__":
    main()
