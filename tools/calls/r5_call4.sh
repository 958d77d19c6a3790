# round 5 call 4: the whole GPU suite as the driver runs it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c4; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu.txt | tail -20
