R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c29
mkdir -p $O
cd $R
for i in 1 2 3; do
timeout 120 python tools/wino_probe.py --res 0 --iters 2 --rounds 1 --shape 64,32,32,96,96 --shape 64,64,64,48,48 --shape 64,32,32,96,192 --wino 70 2>&1 | grep "wino70: max"
done
