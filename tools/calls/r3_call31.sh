R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python tools/find_copies.py 2>&1 | tail -32
