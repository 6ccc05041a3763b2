# usage: tools/abn.sh "variantA variantB ..." [rounds] [bench args]  -- alternating 20-step bench runs on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 ${2:-3}); do for v in $1; do
GBN_AMD_LIB=variants/libgblastn_amd_$v.so timeout 600 python bench.py --no-cpu-baseline --steps ${STEPS:-20} --no-side-workloads $3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', round(d['ms_per_step'],3), round(d['roofline']['scan_stage']['avg_ms'],3), [round(x,2) for x in d['roofline']['scan_stage']['avg_ms_by_kernel']], d['config'].get('engine_only',{}).get('ms_per_pass'))"
done; done
