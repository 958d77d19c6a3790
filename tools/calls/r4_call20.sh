cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 600 python bench.py --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one at a time %.0f crops/s %.3f ms | two in flight %s'%(d['value'], d['ms_per_step'], d.get('two_batches_in_flight')))"; done
