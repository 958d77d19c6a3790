# round 5 call 16: final evidence set on HEAD (bench line with live traffic, kernel stats, PMC traffic), small-batch steps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/prof_r5.sh 2>&1 | tail -22
O=gpurun_out/r5c16; mkdir -p $O
bench() { EGONET_AMD_AUTOTUNE=0 EGONET_AMD_GRAPH_MAX_N=$3 timeout 600 python bench.py --batch $1 --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 batch $1: %.0f crops/s %.3f ms/step'%(d['value'], d['ms_per_step']))"; }
for b in 1 4 16 128; do bench $b eager 0; done | tee $O/small_batch_final.txt
for b in 16 64; do bench $b graph 64; done | tee -a $O/small_batch_final.txt
