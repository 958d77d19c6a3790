R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c27
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino4" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 120 python tools/wino_probe.py --shape 3,32,64,48,96 --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,64,64,256,48 --wino 59,70 > $O/probe.txt 2>&1
grep "wino" $O/probe.txt
timeout 600 python tools/f43_bisect.py > $O/bisect.txt 2>&1
grep "F43 on for (every" $O/bisect.txt
timeout 120 python tools/wino4_clk.py 64,64,64,48,48 > $O/wino4_clk.txt 2>&1
grep -E "K loop|item:|epilogue|prologue" $O/wino4_clk.txt
