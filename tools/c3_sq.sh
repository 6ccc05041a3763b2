# SQ counters of the C3 kernels (blastn W=11): instruction counts, wave cycles, LDS.  usage (GPU box): bash tools/c3_sq.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-rXX}; O=$R/gpurun_out/sq_${TAG}_c3; mkdir -p $O/a $O/b
cd $R
ARGS="--workload C3 --steps 2 --warmup 0 --no-cpu-baseline --engine-steps 0 --min-seconds 0 --no-overlap"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/a -- python bench.py $ARGS > /dev/null 2> $O/a.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/b -- python bench.py $ARGS > /dev/null 2> $O/b.err
python tools/prof_summary.py $(find $O/a -name "*.db" | head -1) --counters > $O/sq_counters.csv
python tools/prof_summary.py $(find $O/b -name "*.db" | head -1) --counters | tail -n +2 >> $O/sq_counters.csv
rm -rf $O/a $O/b
grep -E "seed_ext|scan_slice|scan_fold|dynprog_lane|diag_replay|dynprog_wave|seed_ckeys" $O/sq_counters.csv | sort
