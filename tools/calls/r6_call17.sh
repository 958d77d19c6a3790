# round 6 call 17: conv_wino4r_kernel (cfg 92: row-owner waves, one exchange round) -- parity and timing vs cfg 70
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4r" 2>&1 | tail -4
python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,64,64,256,48 --shape 128,64,64,48,48 --shape 16,64,64,48,48 --direct 0 --wino 70,80,92 --iters 20 2>&1 | grep " us \|nan" | grep -v "direct0:"
