R=$GRAFT_REPO_ROOT
cd $R
EGONET_AMD_LANES=0 timeout 600 python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from tools.f43_bisect import run
from egonet_amd import synth
x = synth.synth_crops(64, 3, 256, 256, seed=100).cuda()
base, _ = run(x, {'EGONET_AMD_F43': '0'})
for rep in range(3):
    got, n = run(x, {'EGONET_AMD_F43_MATCH': 'ci96.96_co96.96_k3x3_s1_p1_r0'})
    d = (got - base).abs()
    per_img = d.amax(dim=(1, 2, 3))
    bad = (per_img > 1e-3).nonzero().flatten().tolist()
    print('rep', rep, 'launches', n, 'max', float(d.max()), 'images over 1e-3:', bad, 'elements over 1e-3:', int((d > 1e-3).sum()))
PY
