cd $GRAFT_REPO_ROOT
bash tools/pmc_s2.sh 2>&1 | tail -80
