#!/bin/bash
# FETCH_SIZE of the scan stage's kernels with and without GBN_PROBE_DYN (its own --pmc pass, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-rare_fetch}; mkdir -p $O; cd $R
for v in 0 1; do
  mkdir -p $O/f$v
  GBN_PROBE_DYN=$v timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/f$v -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0 > /dev/null 2> $O/f$v.err
  python tools/prof_summary.py $(find $O/f$v -name "*.db" | head -1) --counters | grep "probe_\|scan_bin" > $O/fetch_dyn$v.csv
  rm -rf $O/f$v; echo "GBN_PROBE_DYN=$v"; cat $O/fetch_dyn$v.csv
done
