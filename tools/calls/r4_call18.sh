R=$GRAFT_REPO_ROOT
cd $R
export EGONET_AMD_LIB=$R/tools/_build/libegonet_hip_probes.so
timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --wino 70,71,72,73,74,75,76,77 --rounds 3 2>&1 | grep "wino.*us "
