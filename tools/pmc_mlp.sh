#!/bin/bash
# SQ wait / issue counters of the MLP kernels (run on the GPU box from the repo root) -> gpurun_out/pmc_mlp/summary.txt
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/pmc_mlp"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm1 /tmp/pm2
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pm1 -- python3 "$root/tools/mlpbench.py" > "$out/p1.log" 2>&1 || { tail -5 "$out/p1.log"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pm2 -- python3 "$root/tools/mlpbench.py" > "$out/p2.log" 2>&1 || { tail -5 "$out/p2.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("/tmp/pm1", "/tmp/pm2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        if not n.startswith("mlp_"):
            continue
        per[n][r["Counter_Name"]] += float(r["Counter_Value"])
        k = (d, r["Dispatch_Id"])
        if k not in seen:
            seen.add(k)
            per[n]["calls_" + d] += 1
            per[n]["us_" + d] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(out + "/summary.txt", "w") as fh:
    for n, c in sorted(per.items(), key=lambda kv: -kv[1]["us_/tmp/pm1"]):
        calls = c["calls_/tmp/pm1"] or 1
        wc = c["SQ_WAVE_CYCLES"] or 1
        fh.write("%-52s calls %4d avg %6.1f us | per wave-cycle: wait_any %.2f wait_inst %.2f active %.2f valu %.2f lds %.2f vmem %.2f | per wave: valu %5.0f lds %4.0f vmrd %3.0f vmwr %3.0f salu %4.0f mfma %4.0f ldsconf/ldsinst %.2f\n" % (
            n[:52], calls, c["us_/tmp/pm1"] / calls, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc,
            c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_LDS"] / wc, c["SQ_ACTIVE_INST_VMEM"] / wc,
            c["SQ_INSTS_VALU"] / max(c["SQ_WAVES"], 1), c["SQ_INSTS_LDS"] / max(c["SQ_WAVES"], 1), c["SQ_INSTS_VMEM_RD"] / max(c["SQ_WAVES"], 1),
            c["SQ_INSTS_VMEM_WR"] / max(c["SQ_WAVES"], 1), c["SQ_INSTS_SALU"] / max(c["SQ_WAVES"], 1), c["SQ_INSTS_MFMA"] / max(c["SQ_WAVES"], 1),
            c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_INSTS_LDS"], 1)))
print(open(out + "/summary.txt").read())
PY
