#!/bin/bash
# main-queue busy / gap sums of one step with and without the geometry branch (rocprofv3 kernel trace + tools/step_timeline.py)
cd /tmp && export TMPDIR=/tmp
root="$GRAFT_REPO_ROOT"
rm -rf /tmp/ga /tmp/gb
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ga -- python3 $root/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $root/gpurun_out/gap_a.log 2>&1
PN2_LAB_FREEZE_GEOMETRY=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gb -- python3 $root/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $root/gpurun_out/gap_b.log 2>&1
python3 $root/tools/step_timeline.py /tmp/ga > $root/gpurun_out/timeline_with_branch.txt 2>&1
python3 $root/tools/step_timeline.py /tmp/gb > $root/gpurun_out/timeline_frozen.txt 2>&1
grep -E "step window|^queue" $root/gpurun_out/timeline_with_branch.txt $root/gpurun_out/timeline_frozen.txt
