R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c12
mkdir -p $O
cd $R
timeout 120 python tools/wino_probe.py --shape 2,16,32,16,48 --shape 3,32,64,48,96 --shape 64,64,64,48,48 --shape 64,32,32,96,96 --wino 59,70 > $O/wino4_probe.txt 2>&1
echo "rc $?"
grep -v "rc -2" $O/wino4_probe.txt
