cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05r; mkdir -p $O/kt
cd $R
A="--no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0"
python bench.py $A --steps 32 --warmup 4 > $O/bench_fresh.json 2> $O/bench_fresh.err
timeout 400 rocprofv3 --kernel-trace -d $O/kt -- python bench.py $A --steps 16 --warmup 2 > $O/bench_fresh_rocprof.json 2> $O/kt.err
python tools/timeline.py $(find $O/kt -name "*.db" | head -1) 700 500 > $O/fresh_timeline.txt
rm -rf $O/kt
python -c "import json,sys; j=json.loads(open('$O/bench_fresh.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline'])"
