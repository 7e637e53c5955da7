cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x -k "mega or g5_params or g6_rip or abi" 2>&1 | tail -3
