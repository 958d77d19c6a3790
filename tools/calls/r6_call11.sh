# round 6 call 11: (a) does the round-5 crash reproduce with the round-5 SOURCES (commit 9906431, checked out under _old/) on
# today's pool?  (b) this round's new tests
export PYTHONFAULTHANDLER=1
( cd _old && timeout 600 python -m pytest tests/test_gpu_autograd.py tests/graph_inproc_case.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -25; echo "old-tree rc=$?" )
echo "=== old tree under rocgdb (only useful if the run above crashed)"
( cd _old && timeout 900 rocgdb -batch -ex "set pagination off" -ex run -ex bt -ex "thread apply all bt 8" --args python -m pytest tests/test_gpu_autograd.py tests/graph_inproc_case.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^\[Thread\|^\[New Thread" | tail -80 )
echo "=== new tests"
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_bench_size.py tests/test_gpu_stress_streams.py tests/test_gpu_train_ops.py -q -m gpu -k "chunk or 520 or runs_out or wino4w or bench_batch_64 or f43_network or 86- or wino4s or statistics" 2>&1 | tail -15
