# round 6 call 2: s_memtime timeline of conv_wino4w_kernel (cfg 87) beside conv_wino4_kernel (cfg 78) on 96 -> 96 @ 32 x 32
export EGONET_AMD_LIB=$PWD/tools/_build/libegonet_hip_probes.so
python tools/wino4_clk.py --cfg=87 64,32,32,96,96
python tools/wino4_clk.py --cfg=78 64,32,32,96,96
