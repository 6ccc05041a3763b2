#!/bin/bash
# the cached pass (bench.py --record-cache on) alone: its bench line, its kernel timeline, the host's trace marks
#   usage (on the GPU box): bash tools/r05_cached_timeline.sh TAG  -> gpurun_out/TAG/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05o}; mkdir -p $O/kt
cd $R
A="--record-cache on --no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0"
python bench.py $A --steps 32 --warmup 4 > $O/bench_cached.json 2> $O/bench_cached.err
python bench.py $A --steps 32 --warmup 4 > $O/bench_cached2.json 2>> $O/bench_cached.err
GBN_TRACE=1 python bench.py $A --steps 6 --warmup 2 > /dev/null 2> $O/trace_marks.txt
timeout 400 rocprofv3 --kernel-trace -d $O/kt -- python bench.py $A --steps 16 --warmup 2 > $O/bench_cached_rocprof.json 2> $O/kt.err
python tools/timeline.py $(find $O/kt -name "*.db" | head -1) 700 > $O/cached_timeline.txt
rm -rf $O/kt
for f in $O/bench_cached.json $O/bench_cached2.json; do python -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"; done
