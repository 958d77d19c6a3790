R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c10
mkdir -p $O
cd $R
for d in randn zeros ones; do
echo "== data $d"
timeout 300 python tools/wino_probe.py --data $d --shape 64,64,64,48,48 --shape 64,16,16,192,192 --wino 59,67 --iters 50 2>&1 | grep " us "
done > $O/power_probe.txt
for d in randn zeros; do
echo "== gemm data $d"
timeout 300 python tools/gemm_probe.py --data $d 2>&1 | grep -i "us\|TF" | head -12
done >> $O/power_probe.txt
cat $O/power_probe.txt
