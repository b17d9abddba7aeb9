#!/bin/bash
# A/B of one environment switch on one box: tools/ab_switch.sh PN2_SOMETHING [value] [runs] -- bench.py with the switch at
# `value` (default 0), then unset, alternating; ms per step of each run to gpurun_out/ab/<switch>.log
set -e
sw="$1"; val="${2:-0}"; runs="${3:-3}"
mkdir -p gpurun_out/ab
for i in $(seq 1 "$runs"); do
  env "$sw=$val" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=$val ', j['ms_per_step'])" | tee -a "gpurun_out/ab/$sw.log"
  timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', j['ms_per_step'])" | tee -a "gpurun_out/ab/$sw.log"
done
