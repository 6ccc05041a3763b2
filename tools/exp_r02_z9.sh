#!/bin/bash
mkdir -p gpurun_out/z9
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_definitions.py -q -m gpu -x > gpurun_out/z9/parity.log 2>&1; tail -n 3 gpurun_out/z9/parity.log | cut -c1-300
timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>/dev/null > gpurun_out/z9/c3.json; python -c "
import json; d=json.load(open('gpurun_out/z9/c3.json')); print('C3', round(d['ms_per_step'],2), round(d['value'],1), d['config'].get('stage_ms_per_pass'), d['config'].get('hsps_per_pass'), d['config'].get('init_hits_per_pass'))"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/z9/c2.json; python -c "
import json; d=json.load(open('gpurun_out/z9/c2.json')); print('C2', round(d['ms_per_step'],2), round(d['value'],1), round(d['roofline']['frac'],4), d['roofline']['scan_stage']['avg_ms_by_kernel'], d['config'].get('stage_ms_per_pass'))"
