R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c22
mkdir -p $O
cd $R
timeout 600 python tools/f43_bisect.py > $O/bisect.txt 2>&1
cat $O/bisect.txt | grep -v "^$" | tail -15
