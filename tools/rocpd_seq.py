#!/usr/bin/env python3
"""Dispatch-ordered kernel list (start offset, duration, grid, name) of the tail of a rocprofv3 rocpd database.
    python tools/rocpd_seq.py gpurun_out/prof/x_results.db [count]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
name = "name" if "name" in cols else "kernel_name"
gx = [c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols]
wx = [c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols]
sel = "start, end, %s%s%s" % (name, ", " + gx[0] if gx else "", ", " + wx[0] if wx else "")
rows = list(db.execute("select %s from kernels order by start desc limit %d" % (sel, n)))[::-1]
if not rows:
    sys.exit("no kernels; columns: %s" % cols)
t0 = rows[0][0]
for r in rows:
    nm = re.sub(r"\(anonymous namespace\)::", "", r[2])
    nm = re.sub(r"\(.*", "", nm).replace("void ", "")
    extra = " ".join(str(v) for v in r[3:])
    print("%9.1f %8.1f  %-14s %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, extra, nm[:90]))
