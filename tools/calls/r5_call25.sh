# round 5 call 25: the whole GPU suite on the final code and table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c25; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -X faulthandler -m pytest tests/ -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed|Fatal" $O/pytest_gpu.txt | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
