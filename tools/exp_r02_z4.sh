#!/bin/bash
mkdir -p gpurun_out/z4
GBN_POISON=165 GBN_DIAG_COMPACT_MIN=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu --tb=short > gpurun_out/z4/compact.log 2>&1
grep -v "dist-packages" gpurun_out/z4/compact.log | tail -n 40 | cut -c1-400
GBN_POISON=165 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k full_size_c2 --tb=short > gpurun_out/z4/c2.log 2>&1
grep -v "dist-packages" gpurun_out/z4/c2.log | head -n 40 | cut -c1-400
