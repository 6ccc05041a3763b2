#!/bin/bash
# two ranks on the one GPU of the box: does RCCL connect them?
cd $GRAFT_REPO_ROOT
G=$PWD/tests/golden
mkdir -p /tmp/sh && printf "TITLE three\nDBLIST $G/seqn $G/nt.41646578 $G/seqn\n" > /tmp/sh/three.nal
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from gblastn_amd import api
db = api.BlastDb('/tmp/sh/three')
L = "ACGTRYMKWSBDHVN-"
with open('/tmp/sh/q.fa', 'w') as f:
    for oid in (5, 700, 1500, 2004, 2500, 3999):
        s = db.blastna(oid)[:900]
        f.write(">q%d\n%s\n" % (oid, "".join(L[int(x)] for x in s)))
PY
timeout 120 python -m gblastn_amd.blastn_sharded -db /tmp/sh/three -query /tmp/sh/q.fa -out /tmp/sh/one.tsv -max_target_seqs 5 2>&1 | tail -3
wc -l /tmp/sh/one.tsv
for be in nccl gloo; do
  BATCH_SIZE=2000 timeout 180 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29700 -m gblastn_amd.blastn_sharded -db /tmp/sh/three -query /tmp/sh/q.fa -out /tmp/sh/two_$be.tsv -max_target_seqs 5 -backend $be 2>&1 | tail -12
  echo "== $be rc=$?"; cmp /tmp/sh/one.tsv /tmp/sh/two_$be.tsv && echo SAME_$be
done
head -3 /tmp/sh/one.tsv
