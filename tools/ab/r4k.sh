cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_eft.py -q -x -k "twins" 2>&1 | grep -v "^$" | tail -n 25
