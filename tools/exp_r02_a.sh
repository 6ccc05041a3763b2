# round 2, first GPU batch: baseline on this box, range-size experiment (records in the Infinity Cache?), probe U, LDS micro-benchmark
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
./tools/bin/ldsmb > $O/ldsmb.txt 2>&1
summ() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); r=d['roofline']; s=r['scan_stage']; n=r['launches']/d['steps']
    print('$1', 'ms/step', round(d['ms_per_step'],2), 'launches/step', n, 'kernel ms per step (bin,probe,rare)', [round(x*n,2) for x in s['avg_ms_by_kernel']], 'stage', round(s['avg_ms']*n,2))
"; }
python bench.py --no-cpu-baseline --steps 8 --engine-steps 0 2>$O/base.err | tee $O/base.json | summ base
for mib in 2048 512 128 48 24; do
  GBN_RANGE_MIB=$mib python bench.py --no-cpu-baseline --steps 4 --warmup 1 --engine-steps 0 --no-overlap 2>$O/range$mib.err | tee $O/range$mib.json | summ range$mib
done
python bench.py --no-cpu-baseline --steps 4 --warmup 1 --engine-steps 0 --no-overlap 2>/dev/null | summ base_nooverlap
for v in u1 u4; do
  GBN_AMD_LIB=variants/libgblastn_amd_$v.so python bench.py --no-cpu-baseline --steps 6 --engine-steps 0 2>/dev/null | tee $O/$v.json | summ $v
done
