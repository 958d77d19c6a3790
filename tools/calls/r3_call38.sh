R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c38
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv_fc" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 900 python tools/retune.py --out $O/gfx950.json --match k1x1_s1_p0 > $O/retune.log 2>&1; tail -25 $O/retune.log
