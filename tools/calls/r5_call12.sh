# round 5 call 12: the profiler test + the graphed small-batch test: is the trigger the profiler, and is it specific to conv_pw nodes?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c12; mkdir -p $O
export TMPDIR=/tmp
G="tests/test_gpu_models.py::test_small_batches_replay_their_program_as_a_hipgraph"
P="tests/test_gpu_autograd.py::test_training_loop_runs_no_foreign_conv_kernels"
timeout 600 python -X faulthandler -m pytest $P $G -q -m gpu > $O/a_default.txt 2>&1; echo "profiler + graph, default: $(grep -E 'passed|failed|Fatal' $O/a_default.txt | tail -1 | cut -c1-100)"
EGONET_AMD_PW_FUSE=0 timeout 600 python -X faulthandler -m pytest $P $G -q -m gpu > $O/b_nopw.txt 2>&1; echo "profiler + graph, PW_FUSE=0: $(grep -E 'passed|failed|Fatal' $O/b_nopw.txt | tail -1 | cut -c1-100)"
timeout 600 python -X faulthandler -m pytest tests/test_gpu_autograd.py $G -q -m gpu --deselect $P > $O/c_noprof.txt 2>&1; echo "autograd module without the profiler test + graph: $(grep -E 'passed|failed|Fatal' $O/c_noprof.txt | tail -1 | cut -c1-100)"
