R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c28
mkdir -p $O
cd $R
timeout 600 python tools/f43_bisect.py > $O/bisect.txt 2>&1
grep "F43 on" $O/bisect.txt
EGONET_AMD_LANES=0 timeout 600 python tools/f43_bisect.py 2>&1 | grep "F43 on for (every"
