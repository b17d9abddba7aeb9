#!/bin/bash
# One steady-state step of bench.py as a per-queue kernel timeline (run on the GPU box from the repo root):
#   bash tools/step_timeline.sh [outdir under gpurun_out]      ->  gpurun_out/<outdir>/timeline.txt
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/${1:-tl}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_trace -- python3 "$root/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$out/bench.log" 2>&1 || { tail -5 "$out/bench.log"; exit 1; }
python3 "$root/tools/step_timeline.py" /tmp/tl_trace > "$out/timeline.txt" 2>&1
grep -c . "$out/timeline.txt"
