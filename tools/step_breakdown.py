#!/usr/bin/env python3
"""Steady-state per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fps = [i for i, r in enumerate(rows) if 'fps_kernel<512' in r['Kernel_Name']]
a, b = fps[-8], fps[-3]
nst = 5
sel = rows[a:b]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[r['Kernel_Name']][0] += 1
    agg[r['Kernel_Name']][1] += d
tot = sum(v[1] for v in agg.values())
wall = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 / nst
print("kernel-sum per step %.2f ms; wall per step %.2f ms; kernels/step %d" % (tot / nst / 1e3, wall / 1e3, len(sel) / nst))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-95s n/step=%5.1f avg=%8.1fus tot/step=%7.3fms %5.1f%%" % (n[:95], v[0] / nst, v[1] / v[0], v[1] / nst / 1e3, 100 * v[1] / tot))
