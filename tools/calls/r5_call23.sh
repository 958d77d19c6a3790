# round 5 call 23: conv_s2r with two co-groups per block sharing the gathered tile (Cout % 96 == 0) against one per block
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c23; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "s2r" 2>&1 | tail -2
for b in 1 2; do echo "EGN_S2R_NCGB=$b"; EGN_S2R_NCGB=$b timeout 600 python tools/conv_probe.py --res 0 --iters 20 --rounds 3 --shape 64,64,64,48,96,3,2,1 --shape 64,32,32,48,192,3,2,1 --shape 64,16,16,48,384,3,2,1 --shape 16,64,64,48,96,3,2,1 --cfg 85 2>&1 | grep cfg; done | tee $O/ncgb.txt
