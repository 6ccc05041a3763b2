#!/bin/bash
# Round 6: rocprofv3 evidence for the pass over SORTED records (bench.py --record-cache on): kernel-trace stats, then
# FETCH_SIZE and WRITE_SIZE in passes of their own (never combined with trace domains).
#   usage (on the GPU box): bash tools/r06_cached_profile.sh TAG   -> gpurun_out/prof_TAG/{kernel_stats.csv,pmc.csv,bench_under_rocprof.json}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06c}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O/kt $O/fetch $O/write
cd $R
A="--record-cache on --no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0"
timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --steps 16 --warmup 2 $A > $O/bench_under_rocprof.json 2> $O/kt.err
timeout 500 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- python bench.py --steps 2 --warmup 0 $A > /dev/null 2> $O/fetch.err
timeout 500 rocprofv3 --pmc WRITE_SIZE -d $O/write -- python bench.py --steps 2 --warmup 0 $A > /dev/null 2> $O/write.err
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.csv
python tools/prof_summary.py $(find $O/fetch -name "*.db" | head -1) --counters > $O/pmc.csv
python tools/prof_summary.py $(find $O/write -name "*.db" | head -1) --counters | tail -n +2 >> $O/pmc.csv
rm -rf $O/kt $O/fetch $O/write
head -25 $O/kernel_stats.csv; cat $O/pmc.csv | head -40
