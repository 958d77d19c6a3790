R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c33
mkdir -p $O
cd $R
for i in 1 2 3 4 5; do
timeout 300 python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from tools.f43_bisect import run
from egonet_amd import synth
x = synth.synth_crops(64, 3, 256, 256, seed=100).cuda()
base, _ = run(x, {'EGONET_AMD_F43': '0'})
worst = 0.0
for rep in range(6):
    got, n = run(x, {})
    worst = max(worst, float((got - base).abs().max()))
print('6 multi-stream forwards with F(4x4,3x3) (%d launches each): worst max diff vs F(2x2,3x3)-only %.3e' % (n, worst))
PY
done 2>&1 | grep "worst" | tee $O/f43_stability.txt
