# usage: tools/abv.sh "variantA variantB ..." [rounds]  -- alternating bench runs on one box; prints the step, the scan kernels
# inside the pipeline and alone (engine_only.scan_kernels_ms); environment (GBN_*) is passed through
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 ${2:-2}); do for v in $1; do
GBN_AMD_LIB=variants/libgblastn_amd_$v.so timeout 600 python bench.py --no-cpu-baseline --steps ${STEPS:-20} --no-side-workloads 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', round(d['ms_per_step'],3), round(d['roofline']['scan_stage']['avg_ms'],3), [round(x,2) for x in d['roofline']['scan_stage']['avg_ms_by_kernel']], [round(x,2) for x in d['config'].get('engine_only',{}).get('scan_kernels_ms',[])])"
done; done
