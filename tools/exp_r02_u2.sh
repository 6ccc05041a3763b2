#!/bin/bash
# probe kernel: 16-byte loads per lane and round (GBN_PROBE_U), scan stage alone, same box
for r in 1 2; do
for v in "" u1 u3 u4; do
  if [ -z "$v" ]; then lib=""; name=u2; else lib="variants/libgblastn_amd_$v.so"; name=$v; fi
  echo -n "$name: "; GBN_AMD_LIB=$lib timeout 300 python tools/scan_ablate.py 50000 5000 2>&1 | tail -n 1
done; done
