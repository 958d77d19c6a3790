# round 6 call 6: conv_wino4h_kernel after the 12-bit offset fix -- parity, then the stamp analysis (together / alone)
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4h" 2>&1 | tail -3
export EGONET_AMD_LIB=$PWD/tools/_build/libegonet_hip_probes.so
for sk in 0 250; do EGN_W4H_SKEW=$sk python tools/wino4_clk.py --cfg=89 64,64,64,48,48; done
