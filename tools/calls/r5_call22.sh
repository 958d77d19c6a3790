# round 5 call 22: cfg 85 into the tile table (the 40 stride-2 shapes that start at 48 channels, incumbents re-timed in the same
# session), bench before / after, per-class table, parity of the 64-crop forward and the reference fixtures
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c22; mkdir -p $O
export TMPDIR=/tmp
bench() { timeout 400 python bench.py --no-train --no-cpu-baseline --steps 50 --profile-json $O/classes_$1.json 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.0f crops/s %.3f ms/step; kernel sum %.3f ms over %d launches'%(d['value'], d['ms_per_step'], d['backbone']['ms_sum_of_kernels'], d['backbone']['launches']))"; }
bench before | tee $O/bench.txt
timeout 900 python tools/retune.py --out $O/gfx950.json --match _k3x3_s2_,ci48.48_ --retime 7,8,9,17,18 > $O/retune.log 2>&1; tail -45 $O/retune.log
cp $O/gfx950.json egonet_amd/tuned/gfx950.json
bench after | tee -a $O/bench.txt
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "w48" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_bench_size.py -q -m gpu -k "batch_sizes" -s 2>&1 | grep -E "arg-max|passed|failed"
