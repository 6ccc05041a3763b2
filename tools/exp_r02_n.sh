#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload C3 --steps 4 --warmup 1 --no-cpu-baseline --engine-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap', d['ms_per_step'], d['config']['stage_ms_per_pass'], d['config']['hsps_per_pass'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_lane; mkdir -p $O/kt; cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload C3 --steps 2 --warmup 1 --no-cpu-baseline --engine-steps 0 --no-overlap > $O/bench.json 2> $O/kt.err
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.csv
rm -rf $O/kt; head -14 $O/kernel_stats.csv
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['stage_ms_per_pass'], d['config']['hsps_per_pass'])"
