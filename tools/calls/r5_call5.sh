# round 5 call 5: conv_pw.hip (layer1's 1x1 pair): kernel parity, the engine's fused layer1, bench A/B, per-class table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c5; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pw_pair" > $O/pytest_pw.txt 2>&1; tail -25 $O/pytest_pw.txt
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -k "layer1 or hipgraph or eval_mode" > $O/pytest_models.txt 2>&1; tail -15 $O/pytest_models.txt
for f in 0 1; do echo "PW_FUSE=$f"; EGONET_AMD_PW_FUSE=$f timeout 400 python bench.py --no-train --no-cpu-baseline --steps 30 --profile-json $O/classes_$f.json 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f crops/s %.3f ms/step; kernel sum %.3f ms over %d launches'%(d['value'], d['ms_per_step'], d['backbone']['ms_sum_of_kernels'], d['backbone']['launches']))
for k in d['kernels']:
    if '64->' in k['name'] or '256->' in k['name'] or 'pw' in k['name']: print('   %-34s %3d x %7.1f us'%(k['name'], k['launches'], k['avg_us']))
"; done | tee $O/bench_ab.txt
