R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c11
mkdir -p $O
cd $R
timeout 300 python tools/wino_clk.py 64,64,64,48,48 64,16,16,192,192 > $O/wino_clk_kq2.txt 2>&1
cat $O/wino_clk_kq2.txt | grep -v "^cfg 58" 
