# cfg 83 as ONE launch in programs (ticket words): parity; bench A/B: table | 83 (co-tiles on the XCD axis) | 83 (regions on it)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c25; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4c" 2>&1 | tail -6
bench() { timeout 600 python bench.py --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.0f crops/s %.3f ms'%(d['value'], d['ms_per_step']))"; }
bench table
cp egonet_amd/tuned/gfx950.json /tmp/table.json
python - <<'PY'
import json
p='egonet_amd/tuned/gfx950.json'
t=json.load(open(p))
for k in t:
    if k.startswith('n64_h8_w8_ci384.384_co384.384_k3x3_s1_p1'): t[k]['cfg']=83
json.dump(t,open(p,'w'))
PY
bench cfg83-cotile-xcd
EGONET_AMD_LIB=tools/_build/libegonet_hip_nocox.so bench cfg83-region-xcd
bench cfg83-cotile-xcd
EGONET_AMD_LIB=tools/_build/libegonet_hip_nocox.so bench cfg83-region-xcd
timeout 300 python bench.py --no-train --no-cpu-baseline --steps 5 --profile-json $O/profile83.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4c25/profile83.json'))
for r in (d if isinstance(d, list) else d["classes"])[:5]: print(r['name'], r['launches'], '%.1f us'%r['avg_us'])
PY
cp /tmp/table.json egonet_amd/tuned/gfx950.json
