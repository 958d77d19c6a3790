# round 5 call 20: conv_s2r_kernel (cfg 85): parity, then time against the table's choice on the 48-channel stride-2 shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c20; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "s2r" > $O/pytest_s2r.txt 2>&1; tail -12 $O/pytest_s2r.txt | cut -c1-200
timeout 600 python tools/conv_probe.py --res 0 --iters 20 --rounds 3 --shape 64,64,64,48,48,3,2,1 --cfg 8,18,85 --shape 64,64,64,48,96,3,2,1 --cfg 7,17,85 2>&1 | tail -8 | tee $O/probe1.txt
timeout 600 python tools/conv_probe.py --res 0 --iters 20 --rounds 3 --shape 64,32,32,48,192,3,2,1 --cfg 17,85 --shape 64,16,16,48,384,3,2,1 --cfg 17,85 --shape 64,32,32,48,48,3,2,1 --cfg 18,85 2>&1 | tail -8 | tee $O/probe2.txt
