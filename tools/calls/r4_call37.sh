# last check of the committed tree: smoke + the F(4x4,3x3) kernel tests
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4" 2>&1 | grep -E "passed|failed" | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
