cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
GBN_DP_STATS=1 GBN_AMD_LIB=variants/libgblastn_amd_dps.so python bench.py --workload C3 --no-cpu-baseline --steps 1 --warmup 0 --engine-steps 0 --no-overlap 2>&1 | grep "wave DP" | head -2
bash tools/exp_r02_i.sh 2>&1 | head -6
python bench.py --workload C3 --no-cpu-baseline --steps 8 --engine-steps 0 > $O/c3.json 2>$O/c3.err; python -c "
import json
d=json.loads(open('$O/c3.json').read().strip().splitlines()[-1])
print('c3', d['value'], d['ms_per_step'], d['config']['stage_ms_per_pass'])"
