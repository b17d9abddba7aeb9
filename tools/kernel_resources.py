#!/usr/bin/env python3
"""Registers, spills and occupancy of every kernel of one csrc/*.hip file, from `hipcc -Rpass-analysis=kernel-resource-usage`
(cross-compiles without a GPU; also leaves the gfx950 assembly in --asm FILE).

    python tools/kernel_resources.py pn2_mlp_bwd.hip [--asm /tmp/bwd.s] [--against REV]

--against REV compiles the same file as of a git revision beside it and prints both (a kernel edit that starts to spill, or
loses a wave of occupancy, shows here before it is measured).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "khairil_tum-facade_semantic_segmentation_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only",
         "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(REPO, "include"), "-I", CSRC]
KEYS = (("VGPRs", "v"), ("AGPRs", "a"), ("VGPRs Spill", "spill"), ("ScratchSize [bytes/lane]", "scratch"),
        ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "lds"))


def resources(path, asm):
    out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [path, "-o", asm],
                         capture_output=True, text=True)
    if out.returncode:
        sys.exit(out.stderr[-3000:])
    rows, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = rows.setdefault(m.group(1), {})
            continue
        for key, short in KEYS:
            m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
            if m and cur is not None:
                cur[short] = int(m.group(1))
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
    return {n.replace("(anonymous namespace)::", "").replace("void ", ""): r for n, r in zip(names, rows.values())}


def fmt(r):
    return " ".join("%s=%s" % (s, r.get(s, "-")) for _, s in KEYS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--asm", default=None)
    ap.add_argument("--against", default=None)
    a = ap.parse_args()
    src = a.source if os.path.exists(a.source) else os.path.join(CSRC, a.source)
    tmp = tempfile.mkdtemp()
    new = resources(src, a.asm or os.path.join(tmp, "new.s"))
    old = None
    if a.against:
        rel = os.path.relpath(src, REPO)
        prev = os.path.join(tmp, os.path.basename(src))
        with open(prev, "w") as f:
            f.write(subprocess.run(["git", "show", "%s:%s" % (a.against, rel)], cwd=REPO, capture_output=True, text=True,
                                   check=True).stdout)
        old = resources(prev, os.path.join(tmp, "old.s"))
    for name, r in new.items():
        print("%-72s %s" % (name[:72], fmt(r)))
        if old is not None and old.get(name) != r:
            print("%-72s %s" % ("    " + a.against + ":", fmt(old.get(name, {}))))


if __name__ == "__main__":
    main()
