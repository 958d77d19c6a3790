R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c25
mkdir -p $O
cd $R
for v in 128 256 512; do
echo "== EGN_W4_ABL=$v"
EGN_W4_ABL=$v timeout 300 python tools/f43_bisect.py 2>&1 | grep "F43 on for (every\|r1  "
done > $O/bisect_variants.txt
cat $O/bisect_variants.txt
