#!/bin/bash
mkdir -p gpurun_out/z17
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/z17/full.log 2>&1; tail -n 8 gpurun_out/z17/full.log | cut -c1-300
for v in 1 0 1 0; do
GBN_SPLIT_SEED=$v timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3 split=$v', round(d['ms_per_step'],2), round(d['value'],1), d['config'].get('stage_ms_per_pass'), d['config'].get('hsps_per_pass'), d['config'].get('init_hits_per_pass'))"
done
for v in 1 0; do
GBN_SPLIT_SEED=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C2 split=$v', round(d['ms_per_step'],2), round(d['value'],1), round(d['roofline']['frac'],4), d['config'].get('hsps_per_pass'))"
done
