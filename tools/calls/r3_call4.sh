R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c4
mkdir -p $O
cd $R
timeout 600 python tools/wino_probe.py --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,16,16,192,192 --shape 64,64,64,64,64 --shape 32,64,64,48,48 --shape 3,24,40,32,96 --direct 44,16,16,0,44,0 --wino 51,59 > $O/wino9_probe.txt 2>&1
timeout 300 python tools/wino_probe.py --shape 64,8,8,384,384 --shape 32,8,8,384,384 --shape 32,16,16,192,192 --direct 16,16,16 --wino 56,61,57,62 >> $O/wino9_probe.txt 2>&1
timeout 300 python tools/wino_clk.py > $O/wino_clk.txt 2>&1
grep "us \|nan = [1-9]" $O/wino9_probe.txt; cat $O/wino_clk.txt
