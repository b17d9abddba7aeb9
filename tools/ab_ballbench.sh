#!/bin/bash
# planned-query / operator timings of two library builds alternating on one box (tools/ab_lib.sh prepare first)
set -euo pipefail
pkg=khairil_tum-facade_semantic_segmentation_amd
keep=$(mktemp); cp "$pkg/libpn2hip.so" "$keep"; trap 'cp "$keep" "$pkg/libpn2hip.so"; rm -f "$keep"' EXIT
for i in $(seq 1 "${1:-3}"); do
  for w in prev new; do
    cp "tools/ab_libs/libpn2hip_$w.so" "$pkg/libpn2hip.so"
    python3 tools/ballbench.py 50 2>/dev/null | grep -E "plan\+query|grid3? +fused" | sed "s/^/$w /"
  done
done
