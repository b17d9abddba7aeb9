#!/usr/bin/env python3
"""profiles/r02/ from the outputs of tools/profile_ball.sh (gpurun_out/prof_r02/): copies the rocprofv3 summaries and
writes ball_query_pmc.json (kernel durations, FETCH/WRITE traffic with the gfx950 correction, SQ counters, and the
sha256 of the kernel sources they were measured on -- bench.py quotes `traffic` only while that hash matches)."""
import csv
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

src, dst = os.path.join(REPO, "gpurun_out", "prof_r02"), os.path.join(REPO, "profiles", "r02")
os.makedirs(dst, exist_ok=True)
for f in ("bench_kernel_stats.csv", "ball_cube_kernel_stats.csv", "ball_facade_kernel_stats.csv", "control_kernel_stats.csv",
          "pmc_fetch_counter_collection.csv", "pmc_write_counter_collection.csv", "pmc_sq1_counter_collection.csv",
          "pmc_sq2_counter_collection.csv"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))


def counters(name):
    out = {}
    for r in csv.DictReader(open(os.path.join(src, name + "_counter_collection.csv"))):
        if "ball_query_binned" in r["Kernel_Name"]:
            out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def kstat(name):
    for r in csv.DictReader(open(os.path.join(src, name + "_kernel_stats.csv"))):
        if "ball_query_binned" in r["Name"]:
            return {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                    "max_us": float(r["MaxNs"]) / 1e3}


fetch, write = counters("pmc_fetch")["FETCH_SIZE"], counters("pmc_write")["WRITE_SIZE"]
sq = {**counters("pmc_sq1"), **counters("pmc_sq2")}
traffic = int(2 * fetch * 1024 + write * 1024)
j = {
    "kernel": "ball_query_binned_kernel<128> = the one launch of pn2_ball_query_group_planned (SA1: B=16, N=4096, S=1024, K=32, D=9), "
              "256-thread workgroups, 16 lanes per centroid",
    "command": "tools/profile_ball.sh on the GPU box: rocprofv3 --kernel-trace --stats -- python3 tools/run_ball.py {cube,facade} 50 ; "
               "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/run_ball.py cube 5 ; same with --pmc WRITE_SIZE (separate "
               "passes) ; two SQ passes ; then tools/make_profile_json.py",
    "source_sha256": bench.kernel_source_hash(),
    "kernel_us_rocprof": {"cube": kstat("ball_cube"), "facade": kstat("ball_facade")},
    "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reads exactly 1/2 of wide coalesced reads on gfx950 -> doubled; WRITE_SIZE exact",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": 32702464,
    "traffic_over_algorithmic": traffic / 32702464,
    "traffic_note": "reads = cell-sorted points 64 KB + plan 64 KB + packed rows 256 KB per block (the packed rows are 64-byte records "
                    "for 48 bytes of payload); writes = idx + grouped, exactly the algorithmic 29.36 MB",
    "sq_counters_per_launch": sq,
    "sq_per_wave": {k: sq[k] / 4096.0 for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if k in sq},
}
json.dump(j, open(os.path.join(dst, "ball_query_pmc.json"), "w"), indent=1)
print(json.dumps({k: j[k] for k in ("source_sha256", "kernel_us_rocprof", "traffic_over_algorithmic", "sq_per_wave")}, indent=1))
