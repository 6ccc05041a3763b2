# usage: tools/abn.sh "v1 v2 ..." [rounds] -- bench runs of several variants on one box, per-kernel ms from the diagnostics
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${2:-2}); do for v in $1; do
GBN_AMD_LIB=variants/libgblastn_amd_$v.so python bench.py --no-cpu-baseline --steps 6 --engine-steps 0 ${BENCH_ARGS} | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline'].get('scan_stage',{}); print('$v', 'ms/pass', round(d['ms_per_step'],2), 'kernels', [round(x,2) for x in s.get('avg_ms_by_kernel',[])])"
done; done
