#!/usr/bin/env python3
"""profiles/<round>/ (PN2_ROUND, default r04) from the outputs of tools/profile_ball.sh (gpurun_out/prof_<round>/): copies the rocprofv3 summaries and
writes ball_query_pmc.json -- for the OPERATOR-LEVEL kernel (ball_query_group_grid_kernel = pn2_ball_query_group, what
bench.py's roofline.frac prices) and for the planned query (ball_query_binned_kernel): kernel durations, FETCH/WRITE
traffic with the gfx950 correction, SQ counters, and the sha256 of the kernel sources they were measured on (bench.py
quotes `traffic` only while that hash matches)."""
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

ROUND = os.environ.get("PN2_ROUND", "r04")
src, dst = os.path.join(REPO, "gpurun_out", "prof_" + ROUND), os.path.join(REPO, "profiles", ROUND)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "*_kernel_stats.csv")) + glob.glob(os.path.join(src, "pmc_*_counter_collection.csv")):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
ALGO = 32702464


def counters(name, kernel):
    out = {}
    for r in csv.DictReader(open(os.path.join(src, name + "_counter_collection.csv"))):
        if kernel in r["Kernel_Name"]:
            out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def kstat(name, kernel):
    for r in csv.DictReader(open(os.path.join(src, name + "_kernel_stats.csv"))):
        if kernel in r["Name"]:
            return {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                    "max_us": float(r["MaxNs"]) / 1e3}


def entry(mode, kernel, what, waves):
    fetch, write = counters("pmc_fetch_" + mode, kernel)["FETCH_SIZE"], counters("pmc_write_" + mode, kernel)["WRITE_SIZE"]
    sq = {**counters("pmc_sq1_" + mode, kernel), **counters("pmc_sq2_" + mode, kernel)}
    traffic = int(2 * fetch * 1024 + write * 1024)
    return {"kernel": what,
            "kernel_us_rocprof": {"cube": kstat("ball_%s_cube" % mode, kernel), "facade": kstat("ball_%s_facade" % mode, kernel)},
            "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "traffic_bytes_per_launch": traffic,
            "traffic_over_algorithmic": traffic / ALGO, "sq_counters_per_launch": sq,
            "sq_per_wave": {k: sq[k] / waves for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")
                            if k in sq}}


op = entry("selfcontained", "ball_query_group_grid_kernel",
           "ball_query_group_grid_kernel = the one launch of pn2_ball_query_group (SA1: B=16, N=4096, S=1024, K=32, D=9): the operator, "
           "from (xyz, new_xyz, feats) to (idx, grouped); 1024-thread workgroups, one per (block, 64 centroids)", 4096.0)
pl = entry("planned", "ball_query_binned_kernel",
           "ball_query_binned_kernel<128> = the one launch of pn2_ball_query_group_planned on a prebuilt plan; 256-thread workgroups, "
           "16 lanes per centroid", 4096.0)
j = {
    "command": "tools/profile_ball.sh on the GPU box: rocprofv3 --kernel-trace --stats -- python3 tools/run_ball.py {cube,facade} 50 "
               "{selfcontained,planned} ; rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/run_ball.py cube 5 <mode> ; same with "
               "--pmc WRITE_SIZE (separate passes) ; two SQ passes ; then tools/make_profile_json.py",
    "source_sha256": bench.kernel_source_hash(),
    "planned_query_source_sha256": bench.kernel_source_hash(("pn2_ball_binned.hip", "pn2_ball_bin.h", "pn2_common.h")),
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reads exactly 1/2 of wide coalesced reads on gfx950 -> doubled; WRITE_SIZE exact",
    "algorithmic_bytes_per_launch": ALGO,
    # top level = the operator-level kernel (what bench.py's roofline.traffic quotes)
    "traffic_bytes_per_launch": op["traffic_bytes_per_launch"],
    "traffic_over_algorithmic": op["traffic_over_algorithmic"],
    "operator_level": op,
    "planned_query": pl,
}
json.dump(j, open(os.path.join(dst, "ball_query_pmc.json"), "w"), indent=1)
print(json.dumps({"source_sha256": j["source_sha256"],
                  "operator": {k: op[k] for k in ("kernel_us_rocprof", "traffic_over_algorithmic", "sq_per_wave")},
                  "planned": {k: pl[k] for k in ("kernel_us_rocprof", "traffic_over_algorithmic", "sq_per_wave")}}, indent=1))
