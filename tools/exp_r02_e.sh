cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
VARIANTS="${AB:-cur cur2 cur cur2}" DBGS="0" bash tools/exp_r02_c.sh 2>&1 | grep -E "scan [0-9]" | cut -c1-200
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
