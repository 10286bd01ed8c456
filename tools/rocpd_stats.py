#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd SQLite output) as a per-kernel CSV.

usage: python tools/rocpd_stats.py <results.db> [out.csv]
Columns follow rocprofv3's kernel_stats.csv: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs.
"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name, calls, tot, round(avg, 1), round(100.0 * tot / total, 3), mn, mx])


if __name__ == "__main__":
    main()
