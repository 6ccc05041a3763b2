# usage: tools/abc.sh "variantA variantB ..." [rounds]  -- as abv.sh, over CACHED records (bench.py --record-cache on): the step,
# the scan kernels [bin probe rare] inside the pipeline and alone, the sort's GPU time; environment (GBN_*) is passed through
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 ${2:-2}); do for v in $1; do
GBN_AMD_LIB=variants/libgblastn_amd_$v.so timeout 600 python bench.py --record-cache on --no-cpu-baseline --steps ${STEPS:-40} --no-side-workloads 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; cp=c['cached_pass']
        print('$v', round(cp['ms_per_step'],3), [round(x,3) for x in cp['scan_kernels_ms']], [round(x,3) for x in c.get('engine_only',{}).get('scan_kernels_ms',[])], cp.get('records',{}).get('sort_gpu_ms'), round(c['config_measured']['ms'],2))"
done; done
