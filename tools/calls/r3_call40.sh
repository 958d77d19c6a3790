R=$GRAFT_REPO_ROOT
cd $R
timeout 120 python tools/wino_probe.py --shape 3,32,64,48,96 --shape 64,64,64,48,48 --shape 64,32,32,96,96 --shape 64,64,64,256,48 --wino 59,70 2>&1 | grep "wino"
timeout 200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino4" 2>&1 | tail -2
timeout 300 python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from tools.f43_bisect import run
from egonet_amd import synth
x = synth.synth_crops(64, 3, 256, 256, seed=100).cuda()
base, _ = run(x, {'EGONET_AMD_F43': '0'})
worst = 0.0
for rep in range(4):
    got, n = run(x, {})
    worst = max(worst, float((got - base).abs().max()))
print('4 multi-stream forwards with F(4x4,3x3) (%d launches each): worst max diff %.3e' % (n, worst))
PY
