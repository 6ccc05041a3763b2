#!/bin/bash
# Collect the rocprofv3 evidence for profiles/: kernel-trace stats of bench.py and the HBM
# traffic counters in their own passes (never combined with trace domains).
#   usage (on the GPU box): bash tools/profile_round.sh TAG     -> gpurun_out/prof_TAG/*.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-rXX}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O/kt $O/fetch $O/write
cd $R
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 16 --warmup 2 --no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0 > $O/bench_under_rocprof.json 2> $O/kt.err
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0 > /dev/null 2> $O/fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/write -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0 > /dev/null 2> $O/write.err
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.csv
python tools/prof_summary.py $(find $O/fetch -name "*.db" | head -1) --counters > $O/pmc.csv
python tools/prof_summary.py $(find $O/write -name "*.db" | head -1) --counters | tail -n +2 >> $O/pmc.csv
rm -rf $O/kt $O/fetch $O/write
cat $O/kernel_stats.csv | head -12; cat $O/pmc.csv | head -12; cat $O/bench_under_rocprof.json
