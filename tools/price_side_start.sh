#!/bin/bash
# Does it matter WHEN the geometry graph runs beside the step?  bench.py ms per step with the host enqueuing the geometry graph the
# given time after the step's graph has started (lab: PN2_LAB_SIDE_DELAY_US, a host busy-wait in SemSegTrainer._enqueue_geometry).
one() { env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in 1 2; do
  for us in 0 200 400 700 1000 1300; do
    echo "geometry graph enqueued $us us into the step:  $(one PN2_LAB_SIDE_DELAY_US=$us)"
  done
done
