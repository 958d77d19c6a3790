R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r4c12; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py tests/test_gpu_train_hrnet.py tests/test_gpu_gemm.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; grep "passed\|failed\|^FAILED" $O/pytest.log | cut -c1-300 | tail -8
timeout 300 python tools/train_hc_bench.py --steps 10 --warmup 3 2>/dev/null | cut -c1-200
timeout 300 python tools/train_bench.py --steps 200 --warmup 20 2>/dev/null | cut -c1-200
