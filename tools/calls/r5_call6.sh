# round 5 call 6: the model / bench-size parity tests with layer1 on conv_pw.hip (scale folded into the 1x1 filters)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c6; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_models.py -q -m gpu > $O/pytest_models.txt 2>&1; grep -E "^FAILED|passed|failed" $O/pytest_models.txt
timeout 1500 python -m pytest tests/test_gpu_bench_size.py -q -m gpu -k "not training" -s > $O/pytest_bench_size.txt 2>&1; grep -E "^FAILED|passed|failed|arg-max" $O/pytest_bench_size.txt | cut -c1-300
