# re-measure the F(4x4,3x3) configurations against the table (tools/retune_f43.py), then the bench on the new table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c27; mkdir -p $O
timeout 900 python tools/retune_f43.py --out $O/gfx950.json > $O/retune.log 2>&1; tail -3 $O/retune.log
grep -c " -> " $O/retune.log; grep "h8_w8_ci384.384_co384" $O/retune.log | cut -c1-170
cp $O/gfx950.json egonet_amd/tuned/gfx950.json
for i in 1 2; do timeout 600 python bench.py --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new table %.0f crops/s %.3f ms'%(d['value'], d['ms_per_step']))"; done
