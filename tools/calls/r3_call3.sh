# round 3, GPU call 3: conv_wino9_kernel (cfg 59-62) vs conv_wino8_kernel (51/52/56/57) -- results and time
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c3
mkdir -p $O
cd $R
timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,16,16,192,192 --shape 64,64,64,64,64 --shape 3,24,40,32,96 --shape 32,64,64,48,48 --shape 32,32,32,96,96 --direct 44,16,16,0,0,44,16 --wino 51,59,57,62 > $O/wino9_probe.txt 2>&1
timeout 300 python tools/wino_probe.py --shape 64,8,8,384,384 --shape 32,8,8,384,384 --shape 32,16,16,192,192 --shape 5,8,8,96,48 --direct 16,16,16,0 --wino 52,60,56,61,57,62 >> $O/wino9_probe.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_bench_size.py -q -s -p no:cacheprovider > $O/pytest.log 2>&1
cat $O/wino9_probe.txt | grep -v "rc -2"; grep -E "passed|failed|BatchNorm|worst|arg-max|Error" $O/pytest.log | tail
