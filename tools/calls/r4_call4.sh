R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c4
mkdir -p $O
cd $R
for pf in 1 0 1 0; do
EGN_W4_PREFETCH=$pf timeout 600 python bench.py --no-train --no-cpu-baseline --steps 30 > $O/bench_pf$pf.json 2> $O/bench_pf$pf.err
python - <<PY
import json
d=json.loads(open('$O/bench_pf$pf.json').read().strip().splitlines()[-1])
print('prefetch $pf: %.0f crops/s %.3f ms; sum of kernels %.3f; '%(d['value'], d['ms_per_step'], d['backbone']['ms_sum_of_kernels']) + ', '.join('%s %.1f'%(k['name'].split()[1], k['avg_us']) for k in d['kernels'][:4]))
PY
done
