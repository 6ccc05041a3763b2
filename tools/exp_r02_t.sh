#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 1 0 1; do
GBN_DEFER_RARE=$d timeout 300 python bench.py --no-cpu-baseline --engine-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$d C2', round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['frac'],3), [round(x,2) for x in d['roofline']['scan_stage']['avg_ms_by_kernel']], round(d['roofline']['box_copy_GBps']))"
done
