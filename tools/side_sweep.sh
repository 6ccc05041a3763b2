# usage: tools/side_sweep.sh  -- C3 with several caps of the gapped grid, C4 with several traceback thread counts (one box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
p() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['ms_per_step'],3), d.get('ms_per_step_minmax'))"; }
for r in 1 2; do
for w in 16 24 32 48; do GBN_GAP_WAVES=$w timeout 300 python bench.py --workload C3 --steps 32 --no-cpu-baseline --engine-steps 0 2>/dev/null | p "C3 gap_waves=$w"; done
for t in 4 8 16; do timeout 300 python bench.py --workload C4 --steps 80 --no-cpu-baseline --engine-steps 0 --trace-threads $t 2>/dev/null | p "C4 trace_threads=$t"; done
done
