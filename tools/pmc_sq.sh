cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcA $R/gpurun_out/pmcB
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/pmcA -- python tools/scan_ablate.py $ABLATE_ARGS > gpurun_out/pmcA/log.txt 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM -d gpurun_out/pmcB -- python tools/scan_ablate.py $ABLATE_ARGS > gpurun_out/pmcB/log.txt 2>&1
for d in pmcA pmcB; do f=$(find gpurun_out/$d -name "*.db" | head -1); echo "== $d $f"; python tools/prof_summary.py $f --counters | grep -E "scan_bin|probe_bin|probe_rare" ; done
