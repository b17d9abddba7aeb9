#!/bin/bash
# A/B of two BUILDS of libpn2hip.so on one box (GPU boxes differ by +-1 %, a run-time switch changes the code around it):
#   here:        tools/ab_lib.sh prepare [rev]     builds <rev> (default HEAD~1... see below) as tools/ab_libs/libpn2hip_prev.so
#                                                  and the working tree as tools/ab_libs/libpn2hip_new.so
#   on the box:  tools/ab_lib.sh run [runs] [bench.py arguments...]   alternates the two libraries under bench.py
# tools/ab_libs/ travels with gpurun but is not tracked.  The package's own library is restored on ANY exit of `run`.
set -euo pipefail
pkg=khairil_tum-facade_semantic_segmentation_amd
libs=tools/ab_libs
cmd="${1:-run}"; shift || true
case "$cmd" in
prepare)
    rev="${1:-HEAD}"
    mkdir -p "$libs"
    tmp=$(mktemp -d)
    trap 'rm -rf "$tmp"' EXIT
    git archive "$rev" "$pkg" include | tar -x -C "$tmp"
    (cd "$tmp" && python3 "$pkg/build.py" >/dev/null)
    cp "$tmp/$pkg/libpn2hip.so" "$libs/libpn2hip_prev.so"
    python3 "$pkg/build.py" >/dev/null
    cp "$pkg/libpn2hip.so" "$libs/libpn2hip_new.so"
    echo "prev = $rev, new = working tree"; ls -la "$libs"
    ;;
run)
    runs="${1:-3}"; shift || true
    extra=("$@")
    for w in prev new; do
        [ -f "$libs/libpn2hip_$w.so" ] || { echo "missing $libs/libpn2hip_$w.so (tools/ab_lib.sh prepare first)" >&2; exit 2; }
    done
    [ -f "$pkg/libpn2hip.so" ] || { echo "missing $pkg/libpn2hip.so" >&2; exit 2; }
    mkdir -p gpurun_out/ab
    keep=$(mktemp)
    cp "$pkg/libpn2hip.so" "$keep"
    trap 'cp "$keep" "$pkg/libpn2hip.so"; rm -f "$keep"' EXIT
    for i in $(seq 1 "$runs"); do
        for w in prev new; do
            cp "$libs/libpn2hip_$w.so" "$pkg/libpn2hip.so"
            cmp -s "$libs/libpn2hip_$w.so" "$pkg/libpn2hip.so"
            out=$(timeout -k 10 300 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 "${extra[@]}" 2>/dev/null | tail -n 1)
            python3 -c "import sys,json; print('$w', json.loads(sys.argv[1])['ms_per_step'])" "$out" | tee -a gpurun_out/ab/lib.log
        done
    done
    ;;
*)
    echo "usage: tools/ab_lib.sh prepare [rev] | run [runs] [bench.py arguments...]" >&2; exit 2
    ;;
esac
