#!/usr/bin/env python3
"""Matrix-pipe busy fraction per kernel from a rocprofv3 PMC pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d DIR -- python3 tools/mlpbench.py
    python tools/mfma_busy.py DIR > profiles/rNN/mlp_mfma_busy.txt
busy = sum over the launch of SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs).  Durations come from the
same (counter-collecting, hence slower) run."""
import collections
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not files:
    sys.exit("no *counter_collection.csv under " + sys.argv[1])
per = collections.defaultdict(lambda: {"n": 0, "us": 0.0, "busy": 0.0, "insts": 0.0})
seen = set()
for r in csv.DictReader(open(files[0])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    name = re.sub(r"\((anonymous namespace|\(anon).*", "", name)
    name = re.sub(r"\(.*", "", name)
    if not name.startswith(("mlp_", "head_")):
        continue
    d = per[name]
    key = r["Dispatch_Id"]
    if key not in seen:
        seen.add(key)
        d["n"] += 1
        d["us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        d["busy"] += float(r["Counter_Value"])
    elif r["Counter_Name"] == "SQ_INSTS_MFMA":
        d["insts"] += float(r["Counter_Value"])
print("# matrix-pipe busy fraction = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (kernel duration * 2.4 GHz * 1024 SIMDs); "
      "fp32 v_mfma_f32_32x32x2_f32 = 64 cycles each")
print("%-60s %6s %9s %10s %12s" % ("kernel", "calls", "avg us", "mfma busy", "mfma insts"))
for name, d in sorted(per.items(), key=lambda kv: -kv[1]["us"]):
    if d["n"] == 0 or d["us"] == 0:
        continue
    busy = d["busy"] / (d["us"] * 1e-6 * 2.4e9 * 1024)
    print("%-60s %6d %9.1f %9.1f%% %12d" % (name[:60], d["n"], d["us"] / d["n"], 100 * busy, d["insts"] / d["n"]))
