#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_definitions.py tests/test_gpu_parity.py -x -q -k "not two_kernel and not deferred" 2>&1 | tail -2
GBN_GAP_LANE=0 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/prof_cur; mkdir -p $O/kt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload C3 --steps 1 --warmup 1 --no-cpu-baseline --engine-steps 0 --no-overlap > $O/bench.json 2> $O/kt.err
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.csv
rm -rf $O/kt; grep "dynprog" $O/kernel_stats.csv
timeout 300 python bench.py --workload C3 --no-cpu-baseline --engine-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3', round(d['ms_per_step'],2), round(d['value'],1), d['config']['hsps_per_pass'])"
