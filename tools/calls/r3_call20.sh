R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c20
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; head -c 300 $O/bench_n1.json; echo
cp $O/bench_n1.json profiles/r3_bench_n1.json
timeout 900 python -m pytest tests/test_gpu_bench_size.py -x -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "arg-max|passed|failed|Error|assert" $O/pytest.log | head
