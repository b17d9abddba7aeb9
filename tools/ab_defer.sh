set -e
mkdir -p gpurun_out/defer
timeout -k 10 900 python -m pytest tests/test_hip_deferred_sums.py tests/test_hip_mlp.py tests/test_hip_trainer.py tests/test_hip_head.py -x -q > gpurun_out/defer/tests.log 2>&1 || { tail -40 gpurun_out/defer/tests.log; exit 1; }
tail -3 gpurun_out/defer/tests.log
for i in 1 2 3; do
  PN2_DEFER_DW=0 timeout -k 10 300 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('immediate', j['ms_per_step'])" | tee -a gpurun_out/defer/ab.log
  timeout -k 10 300 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deferred ', j['ms_per_step'])" | tee -a gpurun_out/defer/ab.log
done
