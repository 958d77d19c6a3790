# round 5 call 10: the whole GPU suite twice on the build that waits for a program's last run before releasing its graph
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c10; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do timeout 2400 python -X faulthandler -m pytest tests/ -q -m gpu > $O/pytest_gpu_$i.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed|Fatal" $O/pytest_gpu_$i.txt | tail -6; done
