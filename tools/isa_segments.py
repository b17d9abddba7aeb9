#!/usr/bin/env python3
"""Static instruction counts of a kernel between its workgroup barriers (a quick map of where a barrier-phased kernel's
instructions are):  python tools/isa_segments.py file.s kernel_name_substring"""
import re
import sys

src = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(\S*%s\S*):[^\n]*\n(.*?)s_endpgm' % re.escape(name), src, re.S | re.M)
if not m:
    sys.exit("kernel not found")
lines = [l.strip() for l in m.group(2).split('\n')]
lines = [l for l in lines if l and not l.startswith((';', '.', '//')) and not l.endswith(':')]


def kind(l):
    op = l.split()[0]
    for p, k in (('v_', 'valu'), ('s_waitcnt', 'wait'), ('s_', 'salu'), ('ds_', 'lds'), ('global_', 'vmem'), ('buffer_', 'vmem'), ('flat_', 'vmem')):
        if op.startswith(p):
            return k
    return 'other'


seg, cur = [], {}
for l in lines:
    if l.startswith('s_barrier'):
        seg.append(cur)
        cur = {}
        continue
    k = kind(l)
    cur[k] = cur.get(k, 0) + 1
seg.append(cur)
for i, c in enumerate(seg):
    print("segment %d: %4d instructions  %s" % (i, sum(c.values()), c))
print("total (static, loops counted once):", sum(sum(c.values()) for c in seg))
