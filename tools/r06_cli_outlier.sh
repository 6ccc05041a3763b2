#!/bin/bash
# runs the documented invocation N times with the host marks on (GBN_TRACE=1) and keeps the trace of the slowest run
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=${GBN_CLI_DB_DIR:-/tmp/gbn_cli_db}
[ -f $D/c2db.nal ] || python tools/make_synth_blastdb.py $D 2>/dev/null
mkdir -p gpurun_out/r06g/tr
for r in $(seq 1 ${1:-10}); do
GBN_TRACE=1 ./gblastn_amd/bin/blastn_prelim -db $D/c2db -query $D/queries.fa -outfmt 6 -use_gpu true -gpu_id 0 -mode 2 -out /tmp/rows.tsv -timing true 2> gpurun_out/r06g/tr/run$r.txt
grep -o '"total_ms": [0-9.]*\|"db_open_upload_ms": [0-9.]*\|"wait_results_ms": [0-9.]*' gpurun_out/r06g/tr/run$r.txt | tr '\n' ' '; echo
done
