R=$GRAFT_REPO_ROOT
cd $R
export EGONET_AMD_LIB=$R/tools/_build/libegonet_hip_probes.so
timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,64,64,256,48 --wino 70,82 --rounds 4 2>&1 | grep "us \|wino82.*max"
timeout 600 python tools/wino_probe.py --shape 64,16,16,192,192 --shape 64,32,32,96,96 --shape 16,64,64,48,48 --wino 80,83 --rounds 4 2>&1 | grep "us \|wino83.*max"
