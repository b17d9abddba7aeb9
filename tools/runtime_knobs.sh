#!/bin/bash
# HIP runtime switches against the default, alternating on one box (bench.py ms per step): which of the runtime's own launch-path
# options matter for a replayed step.   tools/runtime_knobs.sh [runs]
set -u
runs="${1:-2}"
one() { env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq 1 "$runs"); do
  echo "default $(one PN2_NOP=0)"
  for kv in HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=1 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 GPU_MAX_HW_QUEUES=2 GPU_MAX_HW_QUEUES=8; do
    echo "$kv $(one $kv)"
  done
done
