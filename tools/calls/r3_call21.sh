R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c21
mkdir -p $O
cd $R
timeout 300 python tools/wino_probe.py --shape 64,64,64,256,48 --shape 64,64,64,96,48 --shape 32,64,64,256,48 --wino 59,70 --res 0 > $O/probe.txt 2>&1
grep "wino" $O/probe.txt
