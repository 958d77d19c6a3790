R=$GRAFT_REPO_ROOT
cd $R
timeout 230 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_inference_kitti.py tests/test_gpu_metric.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -3
