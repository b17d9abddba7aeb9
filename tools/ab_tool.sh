#!/bin/bash
# Any tool under the two library builds of tools/ab_lib.sh prepare, alternating on one box:
#   tools/ab_tool.sh <runs> <command...>      every output line is prefixed with prev / new
set -euo pipefail
pkg=khairil_tum-facade_semantic_segmentation_amd
runs="$1"; shift
keep=$(mktemp); cp "$pkg/libpn2hip.so" "$keep"; trap 'cp "$keep" "$pkg/libpn2hip.so"; rm -f "$keep"' EXIT
for i in $(seq 1 "$runs"); do
  for w in prev new; do
    cp "tools/ab_libs/libpn2hip_$w.so" "$pkg/libpn2hip.so"
    "$@" 2>/dev/null | sed "s/^/$w /"
  done
done
