#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_definitions.py -x -q -k "ordered_after or two_kernel" 2>&1 | tail -12
GBN_DIAG_COMPACT_MIN=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_traceback_gpu.py tests/test_cli.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --workload C3 --no-cpu-baseline --engine-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3', round(d['ms_per_step'],2), round(d['value'],1), d['config']['hsps_per_pass'], {k:round(v,2) for k,v in d['config']['stage_ms_per_pass'].items()})"
