R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c23
mkdir -p $O
cd $R
EGONET_AMD_LANES=0 timeout 600 python tools/f43_bisect.py > $O/bisect_serial.txt 2>&1
grep "F43" $O/bisect_serial.txt
