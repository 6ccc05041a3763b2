#!/bin/bash
for r in 1 2; do for v in 1 0; do
GBN_HOST_DETACH=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C2 detach=$v', round(d['ms_per_step'],2), round(d['value'],1), round(d['roofline']['frac'],4), d['config'].get('stage_ms_per_pass'))"
done; done
for v in 1 0; do
GBN_HOST_DETACH=$v timeout 300 python bench.py --workload C4 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C4 detach=$v', round(d['ms_per_step'],2), round(d['value'],1))"
done
