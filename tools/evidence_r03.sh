#!/bin/bash
# Round evidence batch (GPU box, repo root): full GPU tests, bench variants -> gpurun_out/evidence/
set -e
out=gpurun_out/evidence; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -30 $out/gpu_tests.log; exit 1; }
tail -2 $out/gpu_tests.log
timeout -k 10 300 python bench.py > $out/bench.json.log 2>/dev/null; echo default done
timeout -k 10 300 python bench.py --kind facade --no-cpu-baseline > $out/bench_facade.json.log 2>/dev/null; echo facade done
timeout -k 10 300 python bench.py --rgb-off --no-cpu-baseline > $out/bench_rgb_off.json.log 2>/dev/null; echo rgb done
timeout -k 10 300 python bench.py --model pointnet_sem_seg --no-cpu-baseline > $out/bench_control.json.log 2>/dev/null; echo control done
timeout -k 10 300 python bench.py --no-graphs --no-cpu-baseline > $out/bench_no_graphs.json.log 2>/dev/null; echo nographs done
timeout -k 10 300 python bench.py --no-prefetch --no-cpu-baseline > $out/bench_no_prefetch.json.log 2>/dev/null; echo noprefetch done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline > $out/bench_under_torchrun_one_rank.json.log 2>/dev/null; echo torchrun done
PN2_FORCE_DP_PATH=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --no-cpu-baseline > $out/bench_under_torchrun_forced_exchange_path.json.log 2>/dev/null; echo forced done
timeout -k 10 300 python bench.py --drop-in --no-cpu-baseline > $out/bench_drop_in.json.log 2>/dev/null; echo dropin done
timeout -k 10 300 python tools/epochbench.py 100 > $out/epochbench.log 2>&1; tail -3 $out/epochbench.log
for f in $out/bench*.json.log; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], j.get('ms_per_step'), j.get('value'))"; done
