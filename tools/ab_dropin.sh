#!/bin/bash
# drop-in step, module graphs on / off, alternating on ONE box:  bash tools/ab_dropin.sh [rounds] [steps]
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
for i in $(seq 1 "${1:-3}"); do
    for g in 1 0; do
        ms=$(PN2_MODULE_GRAPHS=$g python3 bench.py --drop-in --steps "${2:-100}" --warmup 8 2>/dev/null | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
        echo "round $i PN2_MODULE_GRAPHS=$g  $ms ms/step"
    done
done
