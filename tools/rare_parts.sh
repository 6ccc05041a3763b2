# usage: tools/rare_parts.sh "6 8 12 24" [rounds]  -- the rare kernel's parts per segment, alternating bench runs on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 ${2:-2}); do for v in $1; do
GBN_RARE_PARTS=$v timeout 600 python bench.py --no-cpu-baseline --steps ${STEPS:-20} --no-side-workloads 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('parts $v', round(d['ms_per_step'],3), round(d['roofline']['scan_stage']['avg_ms'],3), [round(x,2) for x in d['roofline']['scan_stage']['avg_ms_by_kernel']], d['config'].get('engine_only',{}).get('scan_kernels_ms'))"
done; done
