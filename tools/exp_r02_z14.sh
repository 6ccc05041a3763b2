#!/bin/bash
mkdir -p gpurun_out/z14
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/z14/full.log 2>&1; tail -n 6 gpurun_out/z14/full.log | cut -c1-300
timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>/dev/null > gpurun_out/z14/c3.json; python -c "
import json; d=json.load(open('gpurun_out/z14/c3.json')); print('C3', round(d['ms_per_step'],2), round(d['value'],1), d['config'].get('stage_ms_per_pass'), d['config'].get('hsps_per_pass'), d['config'].get('init_hits_per_pass'), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],3))"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/z14/c2.json; python -c "
import json; d=json.load(open('gpurun_out/z14/c2.json')); print('C2', round(d['ms_per_step'],2), round(d['value'],1), round(d['roofline']['frac'],4), d['config'].get('stage_ms_per_pass'), d['config'].get('hsps_per_pass'))"
timeout 300 python bench.py --workload C4 --no-cpu-baseline 2>/dev/null > gpurun_out/z14/c4.json; python -c "
import json; d=json.load(open('gpurun_out/z14/c4.json')); print('C4', round(d['ms_per_step'],2), round(d['value'],1), {k:v for k,v in d['config'].items() if not isinstance(v,(dict,list))})"
