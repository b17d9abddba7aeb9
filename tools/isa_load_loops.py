#!/usr/bin/env python3
"""Loops in a kernel's gfx950 assembly that load from global memory and wait for the load inside the same short body
(`load; s_waitcnt vmcnt(0); use` per pass = one memory round trip per pass).  Reads `hipcc -S` output
(tools/kernel_resources.py --asm FILE).

    python tools/isa_load_loops.py /tmp/bwd.s [max body lines, default 60]
"""
import re
import subprocess
import sys

path = sys.argv[1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 60
lines = open(path).read().split("\n")
kernel, labels, start = None, {}, 0
names = {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kernel, labels = m.group(1), {}
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
        continue
    m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and kernel:
        a = labels[m.group(1)]
        body = [x for x in lines[a:i] if x.strip() and not x.strip().startswith(";")]
        loads = [x for x in body if re.search(r"\b(global|buffer)_load", x)]
        waits = [x for x in body if "vmcnt(0)" in x]
        mfma = [x for x in body if "v_mfma" in x]
        if loads and waits and len(body) <= limit and not mfma:
            if kernel not in names:
                names[kernel] = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()
            print("%s\n    lines %d-%d: %d instructions, %d loads, %d vmcnt(0) waits" %
                  (names[kernel].replace("(anonymous namespace)::", ""), a + 1, i + 1, len(body), len(loads), len(waits)))
