#!/bin/bash
# scan_slice_kernel: where the time goes (variants: 1 = present positions only counted, 2 = no global atomic)
for v in "" abl1 abl2; do
  if [ -z "$v" ]; then lib=""; name=full; else lib="variants/libgblastn_amd_$v.so"; name=$v; fi
  echo -n "$name: "; TASK=blastn GBN_AMD_LIB=$lib timeout 300 python tools/scan_ablate.py 1000 100 2>&1 | tail -n 1
done
echo -n "partitioned: "; TASK=blastn GBN_SCAN_SLICE=0 timeout 300 python tools/scan_ablate.py 1000 100 2>&1 | tail -n 1
