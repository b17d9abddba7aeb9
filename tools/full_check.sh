set -e
mkdir -p gpurun_out/full
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/full/gpu_tests.log 2>&1 || { tail -30 gpurun_out/full/gpu_tests.log; exit 1; }
tail -3 gpurun_out/full/gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/full/bench.json.log 2>gpurun_out/full/bench.err
tail -c 1500 gpurun_out/full/bench.json.log
timeout -k 10 300 python tools/inferbench.py --end-to-end > gpurun_out/full/inferbench_e2e.log 2>&1 || true
tail -5 gpurun_out/full/inferbench_e2e.log
timeout -k 10 300 python bench.py --drop-in > gpurun_out/full/bench_drop_in.json.log 2>/dev/null || true
tail -c 600 gpurun_out/full/bench_drop_in.json.log
