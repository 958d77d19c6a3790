R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino4 or winograd_kernels" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,16,16,192,192 --shape 16,64,64,48,48 --shape 16,32,32,96,96 --shape 16,16,16,192,192 --shape 32,16,16,192,192 --shape 64,64,64,96,48 --wino 59,62,70,80 > $O/wino4b_probe.txt 2>&1
grep -v "rc -2" $O/wino4b_probe.txt
