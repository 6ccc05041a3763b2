#!/bin/bash
mkdir -p gpurun_out/z10
timeout 900 python -m pytest tests/test_gpu_definitions.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/z10/parity.log 2>&1; tail -n 3 gpurun_out/z10/parity.log | cut -c1-300
for v in 2 1 2 1; do
GBN_SEED_CKEYS=$v timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>/dev/null > gpurun_out/z10/c3_$v.json; python -c "
import json; d=json.load(open('gpurun_out/z10/c3_$v.json')); print('C3 ckeys=$v', round(d['ms_per_step'],2), round(d['value'],1), d['config'].get('stage_ms_per_pass'), d['config'].get('hsps_per_pass'), d['config'].get('init_hits_per_pass'))"
done
