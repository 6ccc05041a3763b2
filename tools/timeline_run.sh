# kernel timeline of a pipelined bench run: BENCH_ARGS="--workload C3" bash tools/timeline_run.sh [rows]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace -d $O -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $O/bench.json 2> $O/err.txt
python tools/timeline.py $(find $O -name "*.db" | head -1) ${1:-70} | grep -v "rocprim\|rocclr" | tail -${1:-70}
