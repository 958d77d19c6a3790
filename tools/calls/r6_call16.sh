# round 6 call 16: native steps tick BatchNorm counters + reset the loss in one launch (no ATen kernel in the lifter step?)
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_hrnet.py tests/test_gpu_trainer.py tests/test_gpu_autograd.py -q -m gpu -x 2>&1 | tail -4
python tools/train_bench.py --steps 50 --warmup 5 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
EGONET_AMD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/l16 -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 50 --warmup 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/l16 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r6_lifter_stats_after_tick.csv \;
find gpurun_out/l16 -name "*kernel_trace.csv" -delete
grep -i "at::native\|copyBuffer\|step_counters" gpurun_out/r6_lifter_stats_after_tick.csv | cut -c1-160
