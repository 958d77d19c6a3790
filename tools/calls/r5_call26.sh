# round 5 call 26: the 64-crop parity test against the final bench line; SQ counters of conv_s2r_kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c26; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bench_size.py -q -m gpu -k "batch_64" -s > $O/pytest_bench64.txt 2>&1; grep -E "^FAILED|passed|failed|arg-max vs" $O/pytest_bench64.txt | cut -c1-300
bash tools/pmc_s2r.sh > $O/pmc_s2r.log 2>&1; cp gpurun_out/pmc_s2r/r5_pmc_sq_s2r.txt $O/ 2>/dev/null; tail -30 $O/pmc_s2r.log | cut -c1-140
