#!/usr/bin/env python3
"""Per-kernel summary (calls, average, share) of a rocprofv3 rocpd database, as CSV.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = list(db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name, name)))
tot = sum(r[2] for r in rows)
out = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for r in rows:
    out.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]])
