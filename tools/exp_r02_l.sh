#!/bin/bash
# C3 shape: how do probe / bin times depend on the number of hits and on the shard size?
cd $GRAFT_REPO_ROOT
for nq in 1 10 100; do TASK=blastn timeout 200 python tools/scan_ablate.py 1000 $nq 2>&1 | tail -2; done
for ns in 500 2000 5000; do TASK=blastn timeout 200 python tools/scan_ablate.py $ns 100 2>&1 | tail -1; done
