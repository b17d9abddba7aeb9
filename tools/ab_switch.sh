#!/bin/bash
# A/B of one environment switch on one box: tools/ab_switch.sh PN2_SOMETHING [runs] -- bench.py with the switch at 0, then unset, alternating
set -e
sw="$1"; runs="${2:-3}"
mkdir -p gpurun_out/ab
for i in $(seq 1 "$runs"); do
  env "$sw=0" timeout -k 10 300 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=0 ', j['ms_per_step'])" | tee -a "gpurun_out/ab/$sw.log"
  timeout -k 10 300 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', j['ms_per_step'])" | tee -a "gpurun_out/ab/$sw.log"
done
