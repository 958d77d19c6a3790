# round 6 call 7: conv_wino4d_kernel (cfg 90: two independent halves in one 12-wave workgroup) -- parity, timing with start skews, timelines
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4h" 2>&1 | tail -3
for sk in 0 60 120 180 250; do
  echo "== EGN_W4H_SKEW=$sk"
  EGN_W4H_SKEW=$sk timeout 120 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,16,16,192,192 --direct 0 --wino 70,80,90 --iters 20 2>&1 | grep " us " | grep -v direct
done
export EGONET_AMD_LIB=$PWD/tools/_build/libegonet_hip_probes.so
for sk in 0 120; do EGN_W4H_SKEW=$sk timeout 120 python tools/wino4_clk.py --cfg=91 64,64,64,48,48; done
