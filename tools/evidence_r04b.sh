out=gpurun_out/evidence_r04b; mkdir -p $out
step() { name="$1"; secs="$2"; shift 2; timeout -k 10 "$secs" "$@" > "$out/$name" 2> "$out/$name.err"; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit $rc; fi; }
step gpu_tests.log 1000 python -m pytest tests -m gpu -q
tail -2 $out/gpu_tests.log
step bench.json.log 300 python bench.py --batch-sweep
step bench_facade.json.log 300 python bench.py --kind facade --no-cpu-baseline
step bench_rgb_off.json.log 300 python bench.py --rgb-off --no-cpu-baseline
step bench_no_graphs.json.log 300 python bench.py --no-graphs --no-cpu-baseline
step bench_no_prefetch.json.log 300 python bench.py --no-prefetch --no-cpu-baseline
step bench_under_torchrun_one_rank.json.log 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline
PN2_FORCE_DP_PATH=1 step bench_under_torchrun_forced_exchange_path.json.log 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --no-cpu-baseline
step epochbench.log 300 python tools/epochbench.py 100
step inferbench_end_to_end.log 400 python tools/inferbench.py --end-to-end 2000000
for f in $out/bench*.json.log; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], j.get('ms_per_step'), j.get('sustained_ms_per_step'), (j.get('roofline') or {}).get('frac'))"; done
