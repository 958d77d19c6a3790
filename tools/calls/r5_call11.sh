# round 5 call 11: which earlier module makes the graphed small-batch test crash (bisect over modules)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c11; mkdir -p $O
export TMPDIR=/tmp
G="tests/test_gpu_models.py::test_small_batches_replay_their_program_as_a_hipgraph"
for m in test_crop test_gpu_autograd test_gpu_distributed test_gpu_gemm test_gpu_inference_kitti test_gpu_bench_size; do
  timeout 900 python -X faulthandler -m pytest tests/$m.py $G -q -m gpu > $O/$m.txt 2>&1
  echo "$m: $(grep -E 'passed|failed|Fatal' $O/$m.txt | tail -1 | cut -c1-120)"
done
