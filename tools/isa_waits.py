#!/usr/bin/env python3
"""Per-kernel count of global stores / loads against s_waitcnt vmcnt(0) in hipcc -S output: a kernel with about one
vmcnt(0) per store is serialising on store acknowledgements (vmcnt counts stores on gfx9).
    python tools/isa_waits.py file.s"""
import re
import sys

name, st = None, None
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, st = m.group(1), dict(store=0, load=0, w0=0, mfma=0)
        continue
    if name is None:
        continue
    if "global_store" in line or "buffer_store" in line:
        st["store"] += 1
    elif "global_load" in line or "buffer_load" in line:
        st["load"] += 1
    elif "vmcnt(0)" in line:
        st["w0"] += 1
    elif "v_mfma" in line:
        st["mfma"] += 1
    elif "s_endpgm" in line:
        rows.append((name, st))
        name = None
for n, st in rows:
    short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", n)[:70]
    print("%-72s stores %4d loads %4d vmcnt(0) %4d mfma %4d" % (short, st["store"], st["load"], st["w0"], st["mfma"]))
