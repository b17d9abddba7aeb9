one() { env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
L="PN2_TUNE_lab_fps_dummy_regs=0 PN2_TUNE_lab_fps_dummy_lds=256 PN2_TUNE_lab_fps_dummy_threads=64"
for i in 1 2; do
  for us in 0 140 280 560 1120 2000; do
    echo "one sleeping wave per workgroup on 16 CUs for $us us:   $(one $L PN2_TUNE_lab_fps_dummy=$us)"
  done
done
