#!/bin/bash
mkdir -p gpurun_out/z19
timeout 900 python -m pytest tests/test_gpu_definitions.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/z19/parity.log 2>&1; tail -n 6 gpurun_out/z19/parity.log | cut -c1-300
for r in 1 2; do
timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3', round(d['ms_per_step'],2), round(d['value'],1), d['config'].get('stage_ms_per_pass'), d['config'].get('hsps_per_pass'), d['config'].get('init_hits_per_pass'))"
done
