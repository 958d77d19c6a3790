R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c35
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv_fc" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 900 python tools/retune.py --out $O/gfx950.json --match h1_w1 > $O/retune.log 2>&1; tail -12 $O/retune.log
