#!/bin/bash
# bench.py --drop-in (ms per step) with / without the garbage collection in front of every module-graph capture (ops.capture_region),
# and with the allocator's cache emptied as well (lab switches), alternating on one box.
one() { env "$@" timeout -k 10 300 python bench.py --drop-in --steps 100 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in 1 2 3; do
  echo "default (garbage collected before every capture)   $(one PN2_NOP=0)"
  echo "PN2_LAB_NO_COLLECT_BEFORE_CAPTURE=1                $(one PN2_LAB_NO_COLLECT_BEFORE_CAPTURE=1)"
  echo "PN2_LAB_EMPTY_CACHE_BEFORE_CAPTURE=1               $(one PN2_LAB_EMPTY_CACHE_BEFORE_CAPTURE=1)"
done
