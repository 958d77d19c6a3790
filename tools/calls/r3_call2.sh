# round 3, GPU call 2: the new GEMM kernels, the Winograd kernel's timeline, the repaired tests, lifter tests
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c2
mkdir -p $O
cd $R
timeout 300 python tools/gemm_probe.py > $O/gemm_probe.txt 2>&1
timeout 300 python tools/wino_clk.py > $O/wino_clk.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_bench_size.py "tests/test_gpu_autograd.py::test_reference_training_loop_on_the_native_tape_hrnet" tests/test_gpu_train.py -q -s -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 300 python tools/train_bench.py --steps 50 > $O/train_lifter.txt 2>&1
cat $O/gemm_probe.txt; cat $O/wino_clk.txt; grep -E "passed|failed|BatchNorm|worst|arg-max" $O/pytest.log | tail; tail -3 $O/train_lifter.txt
