#!/bin/bash
# SQ counters of the gapped kernels on the C3 shape
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_lane_pmc; mkdir -p $O/a; cd $R
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/a -- python bench.py --workload C3 --steps 1 --warmup 0 --no-cpu-baseline --engine-steps 0 --no-overlap > /dev/null 2> $O/a.err
python tools/prof_summary.py $(find $O/a -name "*.db" | head -1) --counters | grep "dynprog\|kernel,counter" 
tail -3 $O/a.err
rm -rf $O/a
