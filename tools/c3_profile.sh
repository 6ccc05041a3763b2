# C3 shape (blastn W=11, 10,000 x 1 kb queries in 100 kb batches vs 5 Gbp): bench line + kernel-trace stats
#   usage (on the GPU box): bash tools/c3_profile.sh TAG   -> gpurun_out/prof_TAG_c3/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-rXX}; O=$R/gpurun_out/prof_${TAG}_c3; mkdir -p $O/kt
cd $R
python bench.py --workload C3 --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null > $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload C3 --steps 4 --warmup 1 --no-cpu-baseline --engine-steps 0 > $O/bench_under_rocprof.json 2> $O/kt.err
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.csv
rm -rf $O/kt
cat $O/bench.json; head -14 $O/kernel_stats.csv
