# conv_wino4bk_kernel (cfg 84): parity, ticket form, stress; re-measure the F(4x4,3x3) configurations; small-batch bench before / after
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c34; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stress_streams.py -q -m gpu -k "wino4bk or ticket or 84-" 2>&1 | tail -4
timeout 900 python tools/retune_f43.py --out $O/gfx950.json > $O/retune.log 2>&1; tail -2 $O/retune.log
grep " 84: " $O/retune.log | awk '$4 == 84' | cut -c1-150
bench() { EGONET_AMD_AUTOTUNE=0 timeout 600 python bench.py --batch $1 --no-train --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 batch $1: %.0f crops/s %.3f ms/step'%(d['value'], d['ms_per_step']))"; }
for b in 1 4 16; do bench $b before; done
cp $O/gfx950.json egonet_amd/tuned/gfx950.json
for b in 1 4 16 64; do bench $b after; done
