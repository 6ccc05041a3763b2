#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_definitions.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_lane; mkdir -p $O/kt; cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload C3 --steps 2 --warmup 1 --no-cpu-baseline --engine-steps 0 --no-overlap > $O/bench.json 2> $O/kt.err
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.csv
rm -rf $O/kt; head -8 $O/kernel_stats.csv; grep "dynprog" $O/kernel_stats.csv
