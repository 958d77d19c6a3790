R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c9
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "winograd" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,16,16,192,192 --shape 64,8,8,384,384 --shape 32,64,64,48,48 --shape 32,32,32,96,96 --shape 32,16,16,192,192 --shape 32,8,8,384,384 --shape 64,64,64,64,64 --wino 59,61,62,67,68 > $O/wino_kq2_probe.txt 2>&1
grep -v "rc -2" $O/wino_kq2_probe.txt
