# item order of the F(4x4,3x3) kernels: co-tiles on the XCD axis from 4 co-tiles up (product build) | from 2 up | only for
# multiples of 8 -- 64-crop bench with the 384-channel 8 x 8 layers on cfg 83
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c26; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4" 2>&1 | tail -4
bench() { timeout 600 python bench.py --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.0f crops/s %.3f ms'%(d['value'], d['ms_per_step']))"; }
python - <<'PY'
import json
p='egonet_amd/tuned/gfx950.json'
t=json.load(open(p))
for k in t:
    if k.startswith('n64_h8_w8_ci384.384_co384.384_k3x3_s1_p1'): t[k]['cfg']=83
json.dump(t,open(p,'w'))
PY
for i in 1 2; do
bench cox-from-4
EGONET_AMD_LIB=tools/_build/libegonet_hip_cox2.so bench cox-from-2
EGONET_AMD_LIB=tools/_build/libegonet_hip_cox99.so bench cox-8-only
done
timeout 300 python bench.py --no-train --no-cpu-baseline --steps 5 --profile-json $O/profile.json > /dev/null 2>&1
EGONET_AMD_LIB=tools/_build/libegonet_hip_cox2.so timeout 300 python bench.py --no-train --no-cpu-baseline --steps 5 --profile-json $O/profile_cox2.json > /dev/null 2>&1
python - <<'PY'
import json
for f in ('profile','profile_cox2'):
    d=json.load(open('gpurun_out/r4c26/%s.json'%f))
    print(f, ' | '.join('%s %.1f'%(r['name'].split()[1], r['avg_us']) for r in (d if isinstance(d, list) else d['classes'])[:4]))
PY
