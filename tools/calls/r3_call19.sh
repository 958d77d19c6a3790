R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c19
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest_gpu.log
