# for the record: the 96 -> 96 @ 32x32 layers on cfg 80 (16 x 16 regions, 512 items) instead of the table's cfg 70, 64-crop bench
cd $GRAFT_REPO_ROOT
bench() { timeout 300 python bench.py --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.0f crops/s %.3f ms'%(d['value'], d['ms_per_step']))"; }
bench table
python - <<'PY'
import json
p='egonet_amd/tuned/gfx950.json'
t=json.load(open(p))
for k in t:
    if k.startswith('n64_h32_w32_ci96.96_co96.96_k3x3_s1_p1'): t[k]['cfg']=80
json.dump(t,open(p,'w'))
PY
bench cfg80-on-96ch
