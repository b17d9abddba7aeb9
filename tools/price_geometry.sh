#!/bin/bash
# What the parts of the geometry graph cost the training step beside them (lab switches: a part is computed once and reused, or
# the sampling kernel replaced by its sleeping footprint -- WRONG results, measurement only): bench.py ms per step, alternating
# on one box.   tools/price_geometry.sh [runs]
set -u
one() { env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq 1 "${1:-2}"); do
  echo "default                                              $(one PN2_NOP=0)"
  echo "no geometry at all (PN2_LAB_FREEZE_GEOMETRY)         $(one PN2_LAB_FREEZE_GEOMETRY=1)"
  echo "geometry graph captured, never replayed              $(one PN2_LAB_NO_SIDE_REPLAY=1)"
  echo "3-NN tables once (PN2_LAB_SKIP_NN=1)                 $(one PN2_LAB_SKIP_NN=1)"
  echo "index inversion once (PN2_LAB_SKIP=inv)              $(one PN2_LAB_SKIP=inv)"
  echo "levels 2-4 once (PN2_LAB_SKIP=deep)                  $(one PN2_LAB_SKIP=deep)"
  echo "row packing + planned query once (PN2_LAB_SKIP=q1)   $(one PN2_LAB_SKIP=q1)"
  echo "sampling kernel: footprint for an instant            $(one PN2_TUNE_lab_fps_dummy=0)"
  echo "sampling kernel: footprint asleep for 560 us         $(one PN2_TUNE_lab_fps_dummy=560)"
  echo "all of the above skipped (layout, plan, packing left) $(one PN2_TUNE_lab_fps_dummy=0 PN2_LAB_SKIP_NN=1 PN2_LAB_SKIP=q1,inv,deep)"
done
