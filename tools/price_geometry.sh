#!/bin/bash
# What the parts of the geometry graph cost the training step beside them (lab switches: the part is computed once and reused,
# WRONG results, measurement only): bench.py ms per step, alternating with the default on one box.
#   tools/price_geometry.sh [runs]
set -u
runs="${1:-2}"
one() { env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq 1 "$runs"); do
  echo "default                              $(one PN2_NOP=0)"
  echo "all of it (PN2_LAB_FREEZE_GEOMETRY)  $(one PN2_LAB_FREEZE_GEOMETRY=1)"
  echo "3-NN tables (PN2_LAB_SKIP_NN=1)      $(one PN2_LAB_SKIP_NN=1)"
  echo "index inversion (PN2_LAB_SKIP=inv)   $(one PN2_LAB_SKIP=inv)"
  echo "levels 2-4 (PN2_LAB_SKIP=deep)       $(one PN2_LAB_SKIP=deep)"
  echo "3-NN + inversion + levels 2-4        $(one PN2_LAB_SKIP_NN=1 PN2_LAB_SKIP=inv,deep)"
done
