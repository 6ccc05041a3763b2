#!/bin/bash
# the documented invocation on the C2 database of bench.py --workload cli (made if it is not there), phases per run, for a few
# settings of the shard upload's thread count
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=${GBN_CLI_DB_DIR:-/tmp/gbn_cli_db}
[ -f $D/c2db.nal ] || python tools/make_synth_blastdb.py $D 2>/dev/null
for t in ${THREADS:-8 12 16 24 8}; do for r in 1 2; do
echo "== GBN_UPLOAD_THREADS=$t run $r"
GBN_UPLOAD_THREADS=$t GBN_TRACE=${TRACE:-0} ./gblastn_amd/bin/blastn_prelim -db $D/c2db -query $D/queries.fa -outfmt 6 -use_gpu true -gpu_id 0 -mode 2 -out /tmp/rows.tsv -timing true 2>&1 | grep -E "timing|scan:|search:" | tail -${TAILN:-1}
done; done
