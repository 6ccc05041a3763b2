#!/bin/bash
# Where do the 1.3-3 s that some runs of the documented invocation take on top come from?  Back-to-back runs as bench_cli.py makes
# them, runs with a pause between them, runs that free their memory in order before they leave, and runs beside a process that
# holds device memory (as the full bench does).  One line per run: total / upload / wait-for-results ms.
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=${GBN_CLI_DB_DIR:-/tmp/gbn_cli_db}
[ -f $D/c2db.nal ] || python tools/make_synth_blastdb.py $D 2>/dev/null
one() { "$@" ./gblastn_amd/bin/blastn_prelim -db $D/c2db -query $D/queries.fa -outfmt 6 -use_gpu true -gpu_id 0 -mode 2 -out /tmp/rows.tsv -timing true 2>&1 | grep -o '"total_ms": [0-9.]*\|"gbn_init_ms": [0-9.]*\|"db_open_upload_ms": [0-9.]*\|"wait_results_ms": [0-9.]*' | tr '\n' ' '; echo; }
N=${1:-6}
one env >/dev/null
echo "== A back to back"; for r in $(seq $N); do one env; done
echo "== B 2 s pause before each"; for r in $(seq $N); do sleep 2; one env; done
echo "== C orderly teardown (GBN_CLI_TEARDOWN=1), back to back"; for r in $(seq $N); do one env GBN_CLI_TEARDOWN=1; done
echo "== D beside a process that holds 80 GB of device memory, back to back"
python - <<'PY' &
import torch, time
x = torch.empty(80 << 30, dtype=torch.uint8, device="cuda"); x.zero_(); torch.cuda.synchronize()
open("/tmp/holder_ready", "w").write("1")
time.sleep(600)
PY
HP=$!
while [ ! -f /tmp/holder_ready ]; do sleep 0.5; done
for r in $(seq $N); do one env; done
echo "== E beside it, 2 s pause before each"; for r in $(seq $N); do sleep 2; one env; done
kill $HP
