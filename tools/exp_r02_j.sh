cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_cli.py tests/test_traceback_gpu.py -x -q 2>&1 | tail -30
