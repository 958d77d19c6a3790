R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c24
mkdir -p $O
cd $R
timeout 120 python tools/wino_probe.py --shape 3,32,64,48,96 --shape 64,64,64,48,48 --shape 64,32,32,96,96 --wino 59,70 > $O/probe.txt 2>&1
grep "wino" $O/probe.txt
timeout 600 python tools/f43_bisect.py > $O/bisect.txt 2>&1
grep "F43" $O/bisect.txt
