#!/bin/bash
mkdir -p gpurun_out/z8
timeout 600 python -m pytest tests/test_chunking.py -q -m gpu -k full_size > gpurun_out/z8/chunk.log 2>&1; tail -n 15 gpurun_out/z8/chunk.log | cut -c1-300
GBN_FUZZ_BASE=5000 GBN_FUZZ_EXTRA=40 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k randomised > gpurun_out/z8/fuzz.log 2>&1; tail -n 5 gpurun_out/z8/fuzz.log | cut -c1-300
GBN_DIAG_COMPACT_MIN=1 GBN_FUZZ_BASE=7000 GBN_FUZZ_EXTRA=30 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k randomised > gpurun_out/z8/fuzz2.log 2>&1; tail -n 5 gpurun_out/z8/fuzz2.log | cut -c1-300
