# conv_wino4c_kernel (cfg 82 / 83): kernel parity tests, then timing against the table's pick on the 384-channel 8 x 8 maps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c22; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4" 2>&1 | tail -15
for n in 64 16 128; do
timeout 300 python tools/wino_probe.py --shape $n,8,8,384,384 --direct 0 --wino 56,57,61,82,83 --iters 50 2>&1 | tail -8
done
timeout 300 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,16,16,192,192 --direct 0 --wino 70,80 --iters 50 2>&1 | tail -8
