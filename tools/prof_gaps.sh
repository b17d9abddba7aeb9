#!/bin/bash
# main-queue busy / gap sums of one step: default, separate geometry graph, main branch alone (rocprofv3 kernel trace + tools/step_timeline.py)
cd /tmp && export TMPDIR=/tmp
root="$GRAFT_REPO_ROOT"
for mode in default separate frozen; do
  rm -rf /tmp/g_$mode
  case $mode in
    default) export PN2_SEPARATE_GEOMETRY_GRAPH=0 PN2_LAB_FREEZE_GEOMETRY=0;;
    separate) export PN2_SEPARATE_GEOMETRY_GRAPH=1 PN2_LAB_FREEZE_GEOMETRY=0;;
    frozen) export PN2_SEPARATE_GEOMETRY_GRAPH=0 PN2_LAB_FREEZE_GEOMETRY=1;;
  esac
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/g_$mode -- python3 $root/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $root/gpurun_out/gap_$mode.log 2>&1
  python3 $root/tools/step_timeline.py /tmp/g_$mode > $root/gpurun_out/timeline_$mode.txt 2>&1
  echo "== $mode"; grep -E "step window|^queue" $root/gpurun_out/timeline_$mode.txt
done
