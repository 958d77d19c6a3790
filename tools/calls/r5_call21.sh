# round 5 call 21: conv_s2r_kernel at 4 waves per SIMD: 4 vs 5 blocks per CU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c21; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "s2r" 2>&1 | tail -2
for b in 4 5; do echo "EGN_S2R_BPC=$b"; EGN_S2R_BPC=$b timeout 600 python tools/conv_probe.py --res 0 --iters 20 --rounds 3 --shape 64,64,64,48,48,3,2,1 --shape 64,64,64,48,96,3,2,1 --shape 64,32,32,48,192,3,2,1 --shape 64,16,16,48,384,3,2,1 --cfg 85 2>&1 | grep cfg; done | tee $O/bpc.txt
