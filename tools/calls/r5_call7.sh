# round 5 call 7: the evidence set (bench line with live traffic, serial kernel stats, PMC traffic, training kernel stats), then
# the 64-crop parity test against the fresh bench line
cd $GRAFT_REPO_ROOT
bash tools/prof_r5.sh 2>&1 | tail -25
cp gpurun_out/prof_r5/bench_n1.json profiles/r5_bench_n1.json
timeout 1500 python -m pytest tests/test_gpu_bench_size.py -q -m gpu -k "batch_64" -s > gpurun_out/prof_r5/pytest_bench64.txt 2>&1; grep -E "^FAILED|passed|failed|arg-max vs" gpurun_out/prof_r5/pytest_bench64.txt | cut -c1-400
