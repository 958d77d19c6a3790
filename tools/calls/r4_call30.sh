# L2 warm-up of the item's filter slice at the item top, with co-tiles on the XCD axis (experiment 2 again, new item order):
# A/B of two library builds, 64-crop bench
cd $GRAFT_REPO_ROOT
EGONET_AMD_LIB=tools/_build/libegonet_hip_warm.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wino4" 2>&1 | tail -2
bench() { timeout 600 python bench.py --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.0f crops/s %.3f ms'%(d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
bench product
EGONET_AMD_LIB=tools/_build/libegonet_hip_warm.so bench warm-up
done
