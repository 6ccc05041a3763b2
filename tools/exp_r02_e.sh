cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
VARIANTS="old cur old cur" DBGS="0" bash tools/exp_r02_c.sh 2>&1 | grep -E "scan [0-9]" | cut -c1-200
VARIANTS="cur" DBGS="2 4 8" bash tools/exp_r02_c.sh 2>&1 | grep -E "scan [0-9]" | cut -c1-200
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
bash tools/abn.sh "old cur" 2 2>&1 | grep ms/pass
