cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
summ() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); r=d['roofline']; s=r['scan_stage']; n=r['launches']/d['steps']
    print('$1', 'ms/step', round(d['ms_per_step'],2), 'kernel ms per step (bin,probe,rare)', [round(x*n,2) for x in s['avg_ms_by_kernel']], 'stage', round(s['avg_ms']*n,2), 'frac', round(r['frac'],3))
"; }
python bench.py --no-cpu-baseline --steps 8 --engine-steps 0 2>$O/new.err | tee $O/new.json | summ new
python bench.py --no-cpu-baseline --steps 8 --engine-steps 0 2>$O/new.err | tee $O/new2.json | summ new2
python bench.py --workload C3 --no-cpu-baseline --steps 4 --warmup 1 --engine-steps 0 2>$O/c3.err | tee $O/c3.json | summ c3
