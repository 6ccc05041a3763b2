#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in cur; do
O=$R/gpurun_out/prof_$v; mkdir -p $O/kt
L=variants/libgblastn_amd_$v.so; [ $v = cur ] && L=gblastn_amd/libgblastn_amd.so
GBN_AMD_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -- python bench.py --workload C3 --steps 1 --warmup 1 --no-cpu-baseline --engine-steps 0 --no-overlap > $O/bench.json 2> $O/kt.err
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.csv
rm -rf $O/kt; echo $v $(grep "dynprog_lane\|dynprog_wave" $O/kernel_stats.csv | cut -d, -f1,4 | tr '\n' ' ')
done
