#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_traceback_gpu.py tests/test_cli.py -x -q 2>&1 | tail -3
for d in 0 1; do
GBN_DEFER_RARE=$d timeout 300 python bench.py --no-cpu-baseline --engine-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$d C2', round(d['ms_per_step'],2), round(d['value']), {k:round(v,2) for k,v in d['config']['stage_ms_per_pass'].items()}, [round(x,2) for x in d['roofline']['scan_stage']['avg_ms_by_kernel']], d['config']['hsps_per_pass'], d['config']['seeds_per_pass'])"
done
