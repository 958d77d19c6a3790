R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c2
mkdir -p $O
cd $R
# 1. cfg 80 measured for every table entry it plans for; 2. the other BASELINE batch sizes; the merged table is used by the rest of the call
timeout 600 python tools/retune.py --out $O/gfx950_retuned.json > $O/retune.log 2>&1; echo "retune rc $?"; tail -12 $O/retune.log
cp $O/gfx950_retuned.json egonet_amd/tuned/gfx950.json
timeout 900 python tools/tune_sizes.py --sizes 1,2,4,8,16,128 --out $O/gfx950.json > $O/tune_sizes.log 2>&1; echo "tune_sizes rc $?"; tail -4 $O/tune_sizes.log
cp $O/gfx950.json egonet_amd/tuned/gfx950.json
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_stress_streams.py -q -m gpu -s > $O/pytest_models_stress.log 2>&1; echo "pytest rc $?"; grep -c "differ from the solo run" $O/pytest_models_stress.log; grep "differ" $O/pytest_models_stress.log | grep -v " 0 of" | head; tail -15 $O/pytest_models_stress.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_bench_size.py -q -m gpu -s -k "forward_and_decode or f43_network" > $O/pytest_bench_size.log 2>&1; echo "pytest rc $?"; grep "arg-max" $O/pytest_bench_size.log | cut -c1-600; tail -3 $O/pytest_bench_size.log | cut -c1-300
timeout 900 python bench.py --no-train --no-cpu-baseline > $O/bench_notrain.json 2> $O/bench_notrain.err; echo "bench rc $?"; python - <<'PY'
import json,os
try:
    d=json.loads(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r4c2/bench_notrain.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['backbone']['ms_sum_of_kernels'])
    for k in d['kernels'][:8]: print(k['name'], k['launches'], k['avg_us'])
except Exception as e: print('bench parse', e)
PY
