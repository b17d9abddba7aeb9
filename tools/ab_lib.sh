#!/bin/bash
# A/B of two builds of libpn2hip.so on ONE GPU box, alternating (box-to-box spread is +-1 %, a change is often smaller).
#   here (no GPU):   bash tools/ab_lib.sh prepare <git-rev>     builds <git-rev> in a worktree -> tools/_ab/base.so, and
#                                                               copies the current library    -> tools/_ab/new.so
#   on the GPU box:  bash tools/ab_lib.sh run [rounds] [bench args...]      (through gpurun: the .so files travel)
# A run-time switch inside one build is NOT an A/B of a kernel change: the switch itself changes the code around it
# (a flag around the epilogue stores cost 80 us per step whichever way it was set).
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
pkg="$root/khairil_tum-facade_semantic_segmentation_amd"
mkdir -p "$root/tools/_ab"
case "$1" in
  prepare)
    rev="${2:?git revision of the baseline}"
    wt="$(mktemp -d /tmp/pn2_ab.XXXXXX)"
    git -C "$root" worktree add --detach "$wt" "$rev" > /dev/null
    (cd "$wt" && python -c "import __graft_entry__ as g; g.build()" > "$wt/build.log" 2>&1) || { tail -5 "$wt/build.log"; exit 1; }
    cp "$wt/khairil_tum-facade_semantic_segmentation_amd/libpn2hip.so" "$root/tools/_ab/base.so"
    cp "$pkg/libpn2hip.so" "$root/tools/_ab/new.so"
    git -C "$root" worktree remove --force "$wt"
    ls -la "$root/tools/_ab"
    ;;
  run)
    rounds="${2:-3}"; shift; shift || true
    cp "$pkg/libpn2hip.so" /tmp/pn2_ab_keep.so
    for i in $(seq "$rounds"); do
      for v in base new; do
        cp "$root/tools/_ab/$v.so" "$pkg/libpn2hip.so"
        timeout -k 10 300 python "$root/bench.py" --no-cpu-baseline --steps 50 "$@" 2>/dev/null |
          python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'], 4))"
      done
    done
    cp /tmp/pn2_ab_keep.so "$pkg/libpn2hip.so"
    ;;
  *) echo "usage: ab_lib.sh prepare <rev> | run [rounds] [bench args]"; exit 2;;
esac
