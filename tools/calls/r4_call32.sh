# the stream-stress log of the final build (45 cases)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c32
timeout 900 python -m pytest tests/test_gpu_stress_streams.py -q -s -m gpu 2>&1 | grep -v "^$" > gpurun_out/r4c32/stress.log; tail -3 gpurun_out/r4c32/stress.log; grep -c "repetitions differ" gpurun_out/r4c32/stress.log
