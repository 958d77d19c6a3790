# round 5 call 17: same-box A/B of the round's two switches (boxes of the pool differ by several percent this round)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c17; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for f in 0 1; do echo -n "PW_FUSE=$f: "; EGONET_AMD_PW_FUSE=$f timeout 400 python bench.py --no-train --no-cpu-baseline --steps 50 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f crops/s %.3f ms/step'%(d['value'], d['ms_per_step']))"; done
for f in 0 all; do echo -n "TRAIN_F43=$f: "; EGONET_AMD_TRAIN_F43=$f timeout 400 python tools/train_hc_bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f crops/s %.2f ms/step'%(d['value'], d['ms_per_step']))"; done
done | tee $O/ab.txt
