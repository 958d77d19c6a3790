# round 6 call 4: conv_wino4h_kernel (cfg 88) -- parity, timing vs cfg 70 / 80 with start skews, stamp timelines
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino4h" 2>&1 | tail -5
for sk in 0 100 200 300 400; do
  echo "== EGN_W4H_SKEW=$sk"
  EGN_W4H_SKEW=$sk python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --direct 0 --wino 70,80,88 --iters 20 2>&1 | grep " us "
done
export EGONET_AMD_LIB=$PWD/tools/_build/libegonet_hip_probes.so
for sk in 0 250; do EGN_W4H_SKEW=$sk python tools/wino4_clk.py --cfg=89 64,64,64,48,48; done
