# round 6 call 1: conv_wino4w_kernel (cfg 86) -- parity test + A/B timing vs cfg 70 / 80 on the 96-channel class
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino4w or wino4b_kernel" 2>&1 | tail -5
python tools/wino_probe.py --shape 64,32,32,96,96 --shape 128,32,32,96,96 --shape 128,16,16,192,192 --shape 64,16,16,192,192 --direct 0 --wino 70,80,86 --iters 20 2>&1 | tail -40
