cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_c3; mkdir -p $O/kt
cd $R
python bench.py --workload C3 --subjects 500 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_pass'], d['config']['init_hits_per_pass'], d['config']['seeds_per_pass'])"
python bench.py --workload C3 --subjects 500 --steps 6 --warmup 2 --no-cpu-baseline --no-overlap 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_pass'])"
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload C3 --subjects 500 --steps 4 --warmup 1 --no-cpu-baseline --no-overlap > $O/bench.json 2> $O/kt.err
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) | head -16
rm -rf $O/kt
