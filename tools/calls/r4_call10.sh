R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r4c10; mkdir -p $O
for f in 0 1; do for at in 1 0; do
EGONET_AMD_GEMM_FUSE=$f EGONET_AMD_AUTOTUNE=$at timeout 300 python -m pytest tests/test_gpu_train.py -q -m gpu -k "ragged" > $O/p_$f_$at.log 2>&1; echo "fuse $f autotune $at rc $? $(grep 'passed\|failed' $O/p_$f_$at.log | tail -1) $(grep 'Max absolute' $O/p_$f_$at.log | head -2 | tr '\n' ' ')"
done; done
