R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c14
mkdir -p $O
cd $R
for r in 1; do
echo "== res $r"
timeout 120 python tools/wino_probe.py --res $r --shape 64,64,64,48,48 --shape 64,32,32,96,96 --wino 59,70,71,73 2>&1 | grep " us \|wino70: max"
done > $O/wino4_abl2.txt
cat $O/wino4_abl2.txt
