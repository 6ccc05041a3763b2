#!/bin/bash
# C3: range size vs the Infinity Cache (tile limit 2^17 = 1 G positions = 250 MB of subject data per range)
cd $GRAFT_REPO_ROOT
for t in 131072 65536 32768 16384; do
GBN_RANGE_TILES=$t timeout 300 python bench.py --workload C3 --steps 8 --warmup 2 --no-cpu-baseline --engine-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles=$t', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['config']['stage_ms_per_pass'].items()}, d['roofline']['launches'])"
done
