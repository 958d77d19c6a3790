R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c7
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep "^FAILED\|^ERROR\|passed\|failed" $O/pytest_gpu.log | cut -c1-300 | tail -30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
