#!/bin/bash
# One steady-state step of `bench.py --drop-in` as a per-queue kernel timeline + per-kernel totals (GPU box, repo root):
#   bash tools/dropin_timeline.sh [outdir under gpurun_out]
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/${1:-dtl}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dtl_trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtl_trace -- python3 "$root/bench.py" --drop-in --steps 8 --warmup 6 > "$out/bench.log" 2>&1 || { tail -5 "$out/bench.log"; exit 1; }
python3 "$root/tools/step_timeline.py" /tmp/dtl_trace 2 'fps_kernel<512' start > "$out/timeline.txt" 2>&1
grep -c . "$out/timeline.txt"
tail -60 "$out/timeline.txt"
