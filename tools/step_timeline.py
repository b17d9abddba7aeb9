#!/usr/bin/env python3
"""Timeline of ONE steady-state training step from a rocprofv3 --kernel-trace CSV of bench.py:
    python tools/step_timeline.py <dir with *_kernel_trace.csv> [step_from_end] [delimiter regex] [start|end]
Per queue: every dispatch with its start offset, duration and the gap since the previous dispatch on that queue ended;
then the sums (busy, gaps) per queue and per kernel name.  The step is delimited by the Adam update launches (adam_step*:
the window runs from one's END to the next one's END) or by another kernel that occurs once per step (drop-in mode with
torch's optimizer: the 4096-point FPS launch, 'fps_kernel<512' start: from one's START to the next one's START)."""
import collections
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
if not files:
    sys.exit("no *kernel_trace.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
delim = re.compile(sys.argv[3] if len(sys.argv) > 3 else 'adam_step')
edge = 'Start_Timestamp' if len(sys.argv) > 4 and sys.argv[4] == 'start' else 'End_Timestamp'
ends = [i for i, r in enumerate(rows) if delim.search(r['Kernel_Name'])]
if len(ends) < back + 1:
    sys.exit("fewer than %d steps in the trace" % (back + 1))
t0 = int(rows[ends[-back - 1]][edge])
t1 = int(rows[ends[-back]][edge])
sel = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    return re.sub(r"\(.*", "", n)[:70]


print("step window %.1f us, %d dispatches, queues: %s" % ((t1 - t0) / 1e3, len(sel), sorted({r[qkey] for r in sel})))
last_end = {}
busy = collections.Counter()
gaps = collections.Counter()
per = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    q = r[qkey]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = max(e, last_end.get(q, 0))
    busy[q] += (e - s) / 1e3
    if gap > 0:
        gaps[q] += gap
    per[(q, short(r['Kernel_Name']))][0] += 1
    per[(q, short(r['Kernel_Name']))][1] += (e - s) / 1e3
    grid = r.get('Grid_Size', r.get('Grid_Size_X', ''))
    wg = r.get('Workgroup_Size', r.get('Workgroup_Size_X', ''))
    print("q%-3s %9.1f %8.1f gap %7.1f  g=%-9s w=%-5s %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, grid, wg, short(r['Kernel_Name'])))
print()
for q in sorted(busy):
    print("queue %s: busy %.1f us, gaps %.1f us" % (q, busy[q], gaps[q]))
print()
for (q, n), v in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("q%-3s %-72s n=%3d tot=%8.1f us avg=%7.1f" % (q, n, v[0], v[1], v[1] / v[0]))
