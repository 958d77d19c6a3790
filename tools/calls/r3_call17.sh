R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c17
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 900 python tools/retune.py --out $O/gfx950.json --match k3x3_s1 --retime 59,62 > $O/retune.log 2>&1; tail -30 $O/retune.log
