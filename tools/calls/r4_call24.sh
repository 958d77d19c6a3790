# cfg 82 / 83 (conv_wino4c_kernel): parity tests; co-tiles vs regions on the XCD axis (A/B of two builds); bench A/B with the
# 384-channel 8 x 8 layers on cfg 83
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c24; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4c" 2>&1 | tail -4
for lib in "" tools/_build/libegonet_hip_nocox.so; do
echo "== lib '$lib'"
for n in 16 64; do
EGONET_AMD_LIB=$lib timeout 300 python tools/wino_probe.py --shape $n,8,8,384,384 --direct 0 --wino 61,82,83 --iters 50 2>&1 | grep "us " | grep -v direct
done
done
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
bench cfg83
bench cfg83
timeout 300 python bench.py --no-train --no-cpu-baseline --steps 5 --profile-json $O/profile83.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4c24/profile83.json'))
rows=d if isinstance(d,list) else d.get('classes', d)
print(str(rows)[:1500])
PY
cp /tmp/table.json egonet_amd/tuned/gfx950.json
bench table
