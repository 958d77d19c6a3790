R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c36
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_train.py tests/test_gpu_kernels.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; head -c 250 $O/bench_n1.json; echo
python - <<'PY'
import json
b=json.load(open('gpurun_out/r3c36/bench_n1.json'))
print('lifter train', b['train_lifter']['ms_per_step'], 'hc', b['train_hc']['ms_per_step'])
PY
