R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c39
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; head -c 250 $O/bench_n1.json; echo
cp $O/bench_n1.json profiles/r3_bench_n1.json
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python - <<'PY'
import json
b=json.load(open('gpurun_out/r3c39/bench_n1.json'))
print('hc', b['train_hc']['ms_per_step'], 'lifter', b['train_lifter']['ms_per_step'])
PY
