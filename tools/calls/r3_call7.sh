R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c7
mkdir -p $O
cd $R
timeout 600 python tools/wino_probe.py --shape 64,32,32,96,96 --shape 64,16,16,192,192 --shape 32,32,32,96,96 --shape 32,16,16,192,192 --shape 5,32,32,96,96 --shape 64,16,16,384,192 --shape 3,16,48,16,48 --direct 16,16,16,16,0,16,0 --wino 59,62,65 > $O/wino43_probe.txt 2>&1
grep -v "rc -2" $O/wino43_probe.txt
