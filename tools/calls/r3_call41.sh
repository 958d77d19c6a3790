R=$GRAFT_REPO_ROOT
cd $R
bash tools/pmc_wino4.sh > /dev/null 2>&1
grep -A8 "wino4" gpurun_out/pmc_wino4/r3_pmc_sq_wino4.txt | grep -E "wino4|CONFLICT|MFMA_BUSY|BUSY_CU|WAIT_INST_LDS"
O=$R/gpurun_out/r3c41
mkdir -p $O
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; head -c 250 $O/bench_n1.json; echo
cp $O/bench_n1.json profiles/r3_bench_n1.json
timeout 900 python -m pytest tests/test_gpu_bench_size.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
