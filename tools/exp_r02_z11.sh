#!/bin/bash
mkdir -p gpurun_out/z11
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_definitions.py -q -m gpu -x > gpurun_out/z11/parity.log 2>&1; tail -n 25 gpurun_out/z11/parity.log | cut -c1-300
for v in 0 1; do
GBN_SCAN_SLICE=$v timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>/dev/null > gpurun_out/z11/c3_$v.json; python -c "
import json; d=json.load(open('gpurun_out/z11/c3_$v.json')); print('C3 slice=$v', round(d['ms_per_step'],2), round(d['value'],1), d['config'].get('stage_ms_per_pass'), d['config'].get('hsps_per_pass'), d['config'].get('init_hits_per_pass'), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],3))"
done
