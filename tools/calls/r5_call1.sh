# round 5 call 1: F(4x4,3x3) in the training tape (stats item end, ex entry point, one-launch packer), GEMM split-K block
# order, K-split wait bound; HC step A/B over EGONET_AMD_TRAIN_F43; lifter A/B over the block order; small-batch probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_gemm.py -q -m gpu 2>&1 | tail -12 | tee $O/pytest_ops.txt
timeout 700 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stress_streams.py -q -m gpu -k "wino4 or ticket" 2>&1 | tail -6 | tee $O/pytest_wino4.txt
for f in 0 fwd all; do echo "TRAIN_F43=$f"; EGONET_AMD_TRAIN_F43=$f timeout 400 python tools/train_hc_bench.py --steps 10 --warmup 3 2>&1 | tail -1; done | tee $O/train_hc_ab.txt
timeout 900 python -m pytest tests/test_gpu_bench_size.py -q -m gpu -k training -s 2>&1 | tail -25 | tee $O/pytest_train32.txt
for r in 0 1; do echo "GEMM_RASTER=$r"; EGONET_AMD_GEMM_RASTER=$r timeout 300 python tools/train_bench.py 2>&1 | tail -1; done | tee $O/train_lifter_ab.txt
timeout 400 python tools/small_batch_probe.py 2>&1 | tail -5 | tee $O/small_batch.txt
timeout 400 python bench.py --no-train --no-cpu-baseline --steps 30 2>&1 | tail -1 | tee $O/bench_infer.json
