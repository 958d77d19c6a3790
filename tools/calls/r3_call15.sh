R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c15
mkdir -p $O
cd $R
timeout 120 python tools/wino4_clk.py > $O/wino4_clk.txt 2>&1
echo "rc $?"
cat $O/wino4_clk.txt
