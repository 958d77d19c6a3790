# the forward bench at the other BASELINE batch sizes (shipped table, autotune off): 1, 4, 16, 128 crops per step
cd $GRAFT_REPO_ROOT
for b in 1 4 16 128; do
EGONET_AMD_AUTOTUNE=0 timeout 600 python bench.py --batch $b --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b: %.0f crops/s %.3f ms/step  dominant %s'%(d['value'], d['ms_per_step'], d['roofline']['kernel'][:40]))"
done
