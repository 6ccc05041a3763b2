#!/bin/bash
# GBN_WAIT_SLEEP=1 (a waiting host thread sleeps through 60 % of what its last wait took) against 0 (spins in the runtime): C4, the
# cached C2 pass, C3 and the headline pass; ms per step and what the CPU quota did
cd ${GRAFT_REPO_ROOT:-/root/repo}
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d.get('ms_per_step_minmax'))"; }
for s in 1 0 1 0; do
echo "== GBN_WAIT_SLEEP=$s"
GBN_WAIT_SLEEP=$s GBN_CPU_ACCOUNT=1 tools/cpu_throttle.sh timeout 250 python bench.py --workload C4 --steps 80 2>/tmp/e.txt | j C4; grep "cpu quota\|search thread\|extension stage" /tmp/e.txt
GBN_WAIT_SLEEP=$s tools/cpu_throttle.sh timeout 250 python bench.py --record-cache on --steps 40 --no-side-workloads --no-cpu-baseline 2>/tmp/e.txt | j cached; grep "cpu quota" /tmp/e.txt
GBN_WAIT_SLEEP=$s timeout 250 python bench.py --workload C3 --steps 16 --no-cpu-baseline 2>/dev/null | j C3
GBN_WAIT_SLEEP=$s timeout 250 python bench.py --steps 20 --no-side-workloads --no-cpu-baseline 2>/dev/null | j headline
done
