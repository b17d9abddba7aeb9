#!/bin/bash
# Which part of the sampling kernel's footprint (16 workgroups on 16 CUs for 0.56 ms) costs the step beside it: the kernel replaced
# by sleeping waves with its threads / registers / LDS, one at a time (lab, wrong results).   tools/price_fps.sh [runs]
one() { env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq 1 "${1:-2}"); do
  echo "default (the sampling kernel at work)                         $(one PN2_NOP=0)"
  echo "no footprint (0 us)                                           $(one PN2_TUNE_lab_fps_dummy=0)"
  echo "560 us: 512 threads, 64 registers, 64 KB LDS (the kernel's)   $(one PN2_TUNE_lab_fps_dummy=560)"
  echo "560 us: 512 threads, 64 registers, no LDS                     $(one PN2_TUNE_lab_fps_dummy=560 PN2_TUNE_lab_fps_dummy_lds=256)"
  echo "560 us: 512 threads, few registers, 64 KB LDS                 $(one PN2_TUNE_lab_fps_dummy=560 PN2_TUNE_lab_fps_dummy_regs=0)"
  echo "560 us: 512 threads, few registers, no LDS                    $(one PN2_TUNE_lab_fps_dummy=560 PN2_TUNE_lab_fps_dummy_regs=0 PN2_TUNE_lab_fps_dummy_lds=256)"
  echo "560 us: 64 threads, few registers, no LDS                     $(one PN2_TUNE_lab_fps_dummy=560 PN2_TUNE_lab_fps_dummy_regs=0 PN2_TUNE_lab_fps_dummy_lds=256 PN2_TUNE_lab_fps_dummy_threads=64)"
done
