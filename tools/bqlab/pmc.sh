#!/bin/bash
# PMC passes over a lab binary (run on the GPU box): tools/bqlab/pmc.sh <out-prefix> <binary> [args...]
# Writes gpurun_out/<prefix>_<pass>.csv (the counter_collection csv of each pass).
cd /tmp && export TMPDIR=/tmp
out="$1"; shift
root="${GRAFT_REPO_ROOT:-/root/repo}"
passes=(
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_BANK_CONFLICT"
 "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_BRANCH"
)
i=0
for p in "${passes[@]}"; do
  d="$root/gpurun_out/pmc_${out}_$i"
  rm -rf "$d"
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d "$d" -- "$@" > "$root/gpurun_out/pmc_${out}_$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$root/gpurun_out/pmc_${out}_$i.log"; }
  f=$(find "$d" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$root/gpurun_out/${out}_pass$i.csv"
  rm -rf "$d"
  i=$((i+1))
done
echo pmc done
