# usage: tools/ab.sh variantA variantB [rounds]  -- alternating bench runs on one box, per-kernel ms from the diagnostics
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${3:-3}); do for v in $1 $2; do
GBN_AMD_LIB=variants/libgblastn_amd_$v.so python bench.py --no-cpu-baseline --steps 6 --engine-steps 0 ${BENCH_ARGS} | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), d['roofline'].get('kernel'), {k:round(v,2) for k,v in d['roofline'].items() if isinstance(v,(int,float))}, d['config'].get('stage_ms_per_pass'))"
done; done
