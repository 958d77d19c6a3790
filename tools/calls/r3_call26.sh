R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c26
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; head -c 300 $O/bench_n1.json; echo
cp $O/bench_n1.json profiles/r3_bench_n1.json
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest_gpu.log
