#!/bin/bash
# ms per step of bench.py under one PN2_TUNE_* value after another: tools/sweep.sh NAME=V NAME=V ... (first a default run)
mkdir -p gpurun_out/ab
run() { env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.4f' % ('$*', j['ms_per_step']))" | tee -a gpurun_out/ab/sweep.log; }
run PN2_NONE=0
for kv in "$@"; do run "$kv"; done
run PN2_NONE=0
