# round 5 call 24: the 40 stride-2 shapes that start at 48 channels re-timed with EVERY configuration in one session (boxes
# differ: stale times of other sessions must not decide), final table; evidence set; small batches; new stress cases
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c24; mkdir -p $O
export TMPDIR=/tmp
ALL=$(python -c "print(','.join(str(i) for i in list(range(1,31))+[85]))")
timeout 900 python tools/retune.py --out $O/gfx950.json --match _k3x3_s2_,ci48.48_ --retime $ALL > $O/retune.log 2>&1; tail -8 $O/retune.log
cp $O/gfx950.json egonet_amd/tuned/gfx950.json
python - <<'PY'
import json
t=json.load(open('egonet_amd/tuned/gfx950.json'))
ks=sorted(k for k in t if '_k3x3_s2_' in k and 'ci48.48_' in k)
print('cfg 85 entries:', sum(1 for k in ks if t[k]['cfg']==85), 'of', len(ks))
for k in ks:
    if k.startswith('n64_') or k.startswith('n32_') or k.startswith('n16_'):
        ms={int(a):b for a,b in t[k]['ms'].items()}
        o=sorted((v,c) for c,v in ms.items() if c!=85)[0]
        print('%-52s cfg %d %.1f us (best other: cfg %d %.1f us)'%(k,t[k]['cfg'],ms[t[k]['cfg']]*1e3,o[1],o[0]*1e3))
PY
timeout 900 python -m pytest tests/test_gpu_stress_streams.py tests/test_gpu_kernels.py -q -m gpu -k "pw_pair or s2r or 85-" 2>&1 | tail -3
bash tools/prof_r5.sh 2>&1 | tail -8
bench() { EGONET_AMD_AUTOTUNE=0 timeout 600 python bench.py --batch $1 --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $1: %.0f crops/s %.3f ms/step'%(d['value'], d['ms_per_step']))"; }
for b in 1 4 16 128; do bench $b; done | tee $O/small_batch.txt
