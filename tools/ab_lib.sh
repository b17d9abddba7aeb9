#!/bin/bash
# A/B of two builds of libpn2hip.so on one box: tools/ab_libs/libpn2hip_prev.so against tools/ab_libs/libpn2hip_new.so (copied over the
# package's library in turn; tools/ab_libs/ travels with gpurun but is not tracked), bench.py alternating [runs] times
runs="${1:-3}"; pkg=khairil_tum-facade_semantic_segmentation_amd; mkdir -p gpurun_out/ab
for i in $(seq 1 "$runs"); do
  for w in prev new; do
    cp tools/ab_libs/libpn2hip_$w.so $pkg/libpn2hip.so
    timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', j['ms_per_step'])" | tee -a gpurun_out/ab/lib.log
  done
done
cp tools/ab_libs/libpn2hip_new.so $pkg/libpn2hip.so
