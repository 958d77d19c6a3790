R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c18
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bench_size.py tests/test_gpu_engine.py tests/test_gpu_golden.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
timeout 600 python bench.py --no-train > $O/bench_n1.json 2> $O/bench_n1.err; head -c 600 $O/bench_n1.json; echo
