#!/bin/bash
# rocprofv3 evidence for the roofline kernel (run on the GPU box from the repo root): kernel-trace stats of bench.py
# and of the ball-query runner, then FETCH_SIZE / WRITE_SIZE in separate PMC passes.  Output: gpurun_out/prof_r02/.
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/prof_r02"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {  # name, "rocprof options", program...
  name="$1"; opts="$2"; shift 2
  rm -rf "/tmp/rp_$name"
  rocprofv3 $opts -d "/tmp/rp_$name" --output-format csv -- "$@" > "$out/$name.log" 2>&1 || { echo "$name failed"; tail -5 "$out/$name.log"; }
  for f in $(find "/tmp/rp_$name" -name "*.csv"); do cp "$f" "$out/${name}_$(basename "$f" | sed 's/^[0-9]*_//')"; done
  rm -rf "/tmp/rp_$name"
  echo "$name done"
}
run bench "--kernel-trace --stats" python3 "$root/bench.py" --steps 10 --warmup 5 --no-cpu-baseline
run ball_cube "--kernel-trace --stats" python3 "$root/tools/run_ball.py" cube 50
run ball_facade "--kernel-trace --stats" python3 "$root/tools/run_ball.py" facade 50
run pmc_fetch "--pmc FETCH_SIZE --kernel-trace" python3 "$root/tools/run_ball.py" cube 5
run pmc_write "--pmc WRITE_SIZE --kernel-trace" python3 "$root/tools/run_ball.py" cube 5
run pmc_sq1 "--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace" python3 "$root/tools/run_ball.py" cube 5
run pmc_sq2 "--pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace" python3 "$root/tools/run_ball.py" cube 5
run control "--kernel-trace --stats" python3 "$root/bench.py" --model pointnet_sem_seg --steps 5 --warmup 3
ls -la "$out"
