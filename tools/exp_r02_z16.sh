#!/bin/bash
mkdir -p gpurun_out/z16
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_definitions.py tests/test_chunking.py tests/test_traceback_gpu.py -q -m gpu -x > gpurun_out/z16/parity.log 2>&1; tail -n 12 gpurun_out/z16/parity.log | cut -c1-300
for r in 1 2; do
timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>/dev/null > gpurun_out/z16/c3.json; python -c "
import json; d=json.load(open('gpurun_out/z16/c3.json')); print('C3', round(d['ms_per_step'],2), round(d['value'],1), d['config'].get('stage_ms_per_pass'), d['config'].get('hsps_per_pass'), d['config'].get('init_hits_per_pass'), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],3))"
done
