#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in sync; do
L=variants/libgblastn_amd_$v.so
GBN_AMD_LIB=$L timeout 300 python bench.py --workload C3 --steps 2 --warmup 0 --no-cpu-baseline --engine-steps 0 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v C3', round(d['ms_per_step'],2), d['config']['hsps_per_pass'], d['config']['init_hits_per_pass'])"
grep dbg /tmp/err.txt | head -12
done
