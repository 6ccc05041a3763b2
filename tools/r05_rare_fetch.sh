#!/bin/bash
# FETCH_SIZE of the scan stage's kernels under an environment switch (its own --pmc pass, no trace domains)
#   usage (on the GPU box): bash tools/r05_rare_fetch.sh OUT VAR a b
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-rare_fetch}; VAR=${2:-GBN_PROBE_DYN}; mkdir -p $O; cd $R
for v in ${3:-0} ${4:-1}; do
  mkdir -p $O/f$v
  env $VAR=$v timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/f$v -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --engine-steps 0 --no-side-workloads --min-seconds 0 > /dev/null 2> $O/f$v.err
  python tools/prof_summary.py $(find $O/f$v -name "*.db" | head -1) --counters | grep "probe_\|scan_bin" > $O/fetch_$v.csv
  rm -rf $O/f$v; echo "$VAR=$v"; cat $O/fetch_$v.csv
done
