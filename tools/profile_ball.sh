#!/bin/bash
# rocprofv3 evidence for the roofline kernels (run on the GPU box from the repo root): kernel-trace stats of bench.py and of
# the ball-query runner (operator-level entry and planned pair, both distributions), then FETCH_SIZE / WRITE_SIZE and SQ
# counters in separate PMC passes.  Output: gpurun_out/prof_<round>/ (PN2_ROUND, default r04; tools/make_profile_json.py turns it
# into profiles/<round>/).
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/prof_${PN2_ROUND:-r04}"
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
for mode in selfcontained planned; do
  run ball_${mode}_cube "--kernel-trace --stats" python3 "$root/tools/run_ball.py" cube 50 $mode
  run ball_${mode}_facade "--kernel-trace --stats" python3 "$root/tools/run_ball.py" facade 50 $mode
  run pmc_fetch_$mode "--pmc FETCH_SIZE --kernel-trace" python3 "$root/tools/run_ball.py" cube 5 $mode
  run pmc_write_$mode "--pmc WRITE_SIZE --kernel-trace" python3 "$root/tools/run_ball.py" cube 5 $mode
  run pmc_sq1_$mode "--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace" python3 "$root/tools/run_ball.py" cube 5 $mode
  run pmc_sq2_$mode "--pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace" python3 "$root/tools/run_ball.py" cube 5 $mode
done
ls "$out" | head -60
