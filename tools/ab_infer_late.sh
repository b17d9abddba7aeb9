for i in 1 2 3; do
  echo "late   $(timeout -k 10 300 python tools/inferbench.py --end-to-end 2000000 2>/dev/null | grep "^end-to-end" | sed 's/.*-> //; s/ blocks.*//' | tr '\n' ' ')"
  echo "ahead  $(PN2_INFER_LATE_SIDE_ENQUEUE=0 timeout -k 10 300 python tools/inferbench.py --end-to-end 2000000 2>/dev/null | grep "^end-to-end" | sed 's/.*-> //; s/ blocks.*//' | tr '\n' ' ')"
done
