R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c13
mkdir -p $O
cd $R
timeout 120 python tools/wino_probe.py --shape 3,32,64,48,96 --shape 64,64,64,48,48 --shape 64,32,32,96,96 --wino 59,70,71,73 > $O/wino4_abl.txt 2>&1
echo "rc $?"
grep " us \|fp64" $O/wino4_abl.txt
