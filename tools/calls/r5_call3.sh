# round 5 call 3: failing model tests in full; training parity at 32 crops with F(4x4,3x3) data gradients on stride-1 layers only
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -x -k "hipgraph" > $O/pytest_graph.txt 2>&1; tail -60 $O/pytest_graph.txt
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --deselect "tests/test_gpu_models.py::test_small_batches_replay_their_program_as_a_hipgraph" > $O/pytest_models.txt 2>&1; grep -E "^FAILED|passed|failed|Error" $O/pytest_models.txt | head
timeout 900 python -m pytest tests/test_gpu_bench_size.py -q -m gpu -k training -s 2>&1 | grep -E "data gradient|worst|launches|passed|failed" | tee $O/train32_dgrad.txt
EGONET_AMD_TRAIN_F43=all timeout 400 python tools/train_hc_bench.py --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-200
