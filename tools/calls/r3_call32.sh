R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c32
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; head -c 250 $O/bench_n1.json; echo
cp $O/bench_n1.json profiles/r3_bench_n1.json
timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
