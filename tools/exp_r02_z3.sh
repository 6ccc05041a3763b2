#!/bin/bash
# every device block filled with a byte pattern before use (GBN_POISON): reads of memory nobody wrote show up
mkdir -p gpurun_out/z3
for v in 165 255 1; do
  GBN_POISON=$v timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/z3/poison_$v.log 2>&1
  echo "poison $v: $(tail -n 1 gpurun_out/z3/poison_$v.log)"
  grep "^FAILED\|^ERROR" gpurun_out/z3/poison_$v.log | head -20
done
