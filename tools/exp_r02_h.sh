cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); r=d['roofline']; s=r['scan_stage']; n=r['launches']/d['steps']
    print('$1', 'ms/step', round(d['ms_per_step'],2), 'launches/step', n, 'scan kernels ms per step', [round(x*n,2) for x in s['avg_ms_by_kernel']], 'stage', round(s['avg_ms']*n,2), d['config']['stage_ms_per_pass'])
"; }
python bench.py --workload C3 --no-cpu-baseline --steps 4 --warmup 1 --engine-steps 0 2>/dev/null | summ part
GBN_SCAN_BINS=1 python bench.py --workload C3 --no-cpu-baseline --steps 4 --warmup 1 --engine-steps 0 2>/dev/null | summ direct
python bench.py --workload C3 --no-cpu-baseline --steps 2 --warmup 1 --engine-steps 0 --no-overlap 2>/dev/null | summ part_noov
GBN_SCAN_BINS=1 python bench.py --workload C3 --no-cpu-baseline --steps 2 --warmup 1 --engine-steps 0 --no-overlap 2>/dev/null | summ direct_noov
