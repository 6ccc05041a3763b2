cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
(time python -m pytest tests/test_gpu_parity.py tests/test_gpu_definitions.py tests/test_workload_size_gpu.py -m gpu -x -q) 2>&1 | tail -5
for rep in 1 2; do
for v in prev base; do
O=/tmp/kt_$v$rep; mkdir -p $O
L=variants/libgblastn_amd_$v.so; [ $v = base ] && L=gblastn_amd/libgblastn_amd.so
GBN_AMD_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d $O -- python bench.py --workload C3 --steps 2 --warmup 0 --no-cpu-baseline --engine-steps 0 --min-seconds 0 --no-overlap > /dev/null 2> $O/err.txt
echo "== $v $(python tools/prof_summary.py $(find $O -name "*.db" | head -1) | grep -E "seed_ex" | tr '\n' ' ')"
GBN_AMD_LIB=$L python bench.py --workload C3 --no-cpu-baseline --engine-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   pass', d['ms_per_step'], d['ms_per_step_minmax'])"
done; done
