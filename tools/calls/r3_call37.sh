R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_r3.sh > gpurun_out/prof_r3_final.log 2>&1
tail -3 gpurun_out/prof_r3_final.log | cut -c1-200
O=$R/gpurun_out/r3c37
mkdir -p $O
cd $R
cp gpurun_out/prof_r3/r3_pmc_traffic.json profiles/r3_pmc_traffic.json
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; head -c 250 $O/bench_n1.json; echo
cp $O/bench_n1.json profiles/r3_bench_n1.json
timeout 300 python tools/frontend_bench.py > $O/frontend.json 2> $O/frontend.err; tail -c 600 $O/frontend.json; echo
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
