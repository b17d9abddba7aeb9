#!/bin/bash
# Round evidence for the training step (run on the GPU box from the repo root):
#   kernel-trace stats of bench.py, one step as a per-queue timeline, the matrix-pipe busy fractions of the MLP kernels.
# Output: gpurun_out/prof_step/.
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/prof_step"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_a /tmp/ps_b /tmp/ps_c
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_a -- python3 "$root/bench.py" --steps 10 --warmup 5 --no-cpu-baseline > "$out/bench_stats.log" 2>&1 || { tail -5 "$out/bench_stats.log"; exit 1; }
cp $(find /tmp/ps_a -name "*kernel_stats.csv") "$out/bench_kernel_stats.csv"
echo "stats done"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps_b -- python3 "$root/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$out/bench_trace.log" 2>&1 || { tail -5 "$out/bench_trace.log"; exit 1; }
python3 "$root/tools/step_timeline.py" /tmp/ps_b > "$out/step_timeline.txt" 2>&1
echo "timeline done"
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/ps_c -- python3 "$root/tools/mlpbench.py" > "$out/mlpbench_pmc.log" 2>&1 || { tail -5 "$out/mlpbench_pmc.log"; exit 1; }
python3 "$root/tools/mfma_busy.py" /tmp/ps_c > "$out/mlp_mfma_busy.txt" 2>&1
echo "mfma done"
timeout -k 10 300 python3 "$root/tools/mlpbench.py" > "$out/mlpbench.log" 2>&1
tail -3 "$out/mlpbench.log"
