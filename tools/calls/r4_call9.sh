R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r4c9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_train.py tests/test_gpu_train_ops.py tests/test_gpu_autograd.py tests/test_gpu_trainer.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; grep "passed\|failed\|^FAILED\|^E " $O/pytest.log | cut -c1-300 | tail -12
for f in 1 0 1 0; do EGONET_AMD_GEMM_FUSE=$f timeout 300 python tools/train_bench.py --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fuse $f:', d.get('ms_per_step'), d.get('value'))
"; done
