#!/bin/bash
# tools/inferbench.py --end-to-end (blocks/s at sub-batches of 32 / 64) with one environment switch set against the default,
# alternating on one box:   tools/ab_infer_switch.sh PN2_SOMETHING [value] [runs]
sw="$1"; val="${2:-0}"
for i in $(seq 1 "${3:-3}"); do
  echo "default  $(timeout -k 10 300 python tools/inferbench.py --end-to-end 2000000 2>/dev/null | grep "^end-to-end" | sed 's/.*-> //; s/ blocks.*//' | tr '\n' ' ')"
  echo "$sw=$val  $(env "$sw=$val" timeout -k 10 300 python tools/inferbench.py --end-to-end 2000000 2>/dev/null | grep "^end-to-end" | sed 's/.*-> //; s/ blocks.*//' | tr '\n' ' ')"
done
