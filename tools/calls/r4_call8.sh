R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r4c8
timeout 900 python -m pytest tests/test_gpu_bench_size.py tests/test_gpu_train_hrnet.py -q -m gpu -k "other_baseline or target_weights" -s > gpurun_out/r4c8/pytest_new.log 2>&1; echo "pytest rc $?"; grep "arg-max\|passed\|failed\|^FAILED\|Error" gpurun_out/r4c8/pytest_new.log | cut -c1-300 | tail
bash tools/prof_r4.sh > gpurun_out/r4c8/prof.log 2>&1; tail -25 gpurun_out/r4c8/prof.log | cut -c1-400
