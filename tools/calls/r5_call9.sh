# round 5 call 9: reproduce the segmentation fault of the graphed small-batch test inside the whole suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c9; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -m gpu -k "pw_pair or hipgraph" > $O/a.txt 2>&1; tail -3 $O/a.txt | cut -c1-200
timeout 900 python -X faulthandler -m pytest tests/test_gpu_inference_kitti.py tests/test_gpu_models.py -q -m gpu -k "kitti or hipgraph" > $O/b.txt 2>&1; tail -3 $O/b.txt | cut -c1-200
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -m gpu > $O/c.txt 2>&1; tail -3 $O/c.txt | cut -c1-200
