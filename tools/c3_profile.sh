# C3 shape (blastn W=11, 10,000 x 1 kb queries in 100 kb batches vs 5 Gbp): bench line, kernel-trace stats and the
# HBM traffic counters in passes of their own (never combined with trace domains)
#   usage (on the GPU box): bash tools/c3_profile.sh TAG   -> gpurun_out/prof_TAG_c3/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-rXX}; O=$R/gpurun_out/prof_${TAG}_c3; mkdir -p $O/kt $O/fetch $O/write
cd $R
timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>/dev/null > $O/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload C3 --steps 4 --warmup 1 --no-cpu-baseline --engine-steps 0 --min-seconds 0 > $O/bench_under_rocprof.json 2> $O/kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- python bench.py --workload C3 --steps 1 --warmup 0 --no-cpu-baseline --engine-steps 0 --min-seconds 0 > /dev/null 2> $O/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -- python bench.py --workload C3 --steps 1 --warmup 0 --no-cpu-baseline --engine-steps 0 --min-seconds 0 > /dev/null 2> $O/write.err
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.csv
python tools/prof_summary.py $(find $O/fetch -name "*.db" | head -1) --counters > $O/pmc.csv
python tools/prof_summary.py $(find $O/write -name "*.db" | head -1) --counters | tail -n +2 >> $O/pmc.csv
rm -rf $O/kt $O/fetch $O/write
cat $O/bench.json; head -16 $O/kernel_stats.csv; grep -v "rocprim\|rocclr\|lut_" $O/pmc.csv | head -24
