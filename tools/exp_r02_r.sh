#!/bin/bash
cd $GRAFT_REPO_ROOT
GBN_DIAG_COMPACT_MIN=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_traceback_gpu.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --workload C3 --no-cpu-baseline --engine-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3', d['ms_per_step'], d['value'], d['config']['stage_ms_per_pass'], d['config']['hsps_per_pass'])"
timeout 300 python bench.py --no-cpu-baseline --engine-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['ms_per_step'], d['value'], d['config']['stage_ms_per_pass'], d['roofline']['scan_stage']['avg_ms_by_kernel'])"
