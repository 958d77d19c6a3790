# round 6 call 18: final evidence on the final sources -- whole GPU suite + smoke, the no-flag bench line, the evidence set again
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py 2>/dev/null | tail -1 > gpurun_out/r6_head_bench_n1.json
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r6_head_bench_n1.json').read())
print('no-flag bench:', b['value'], b['ms_per_step'], 'frac', b['roofline']['frac'], 'traffic src', b['roofline']['traffic_source'][:60], '| hc', b['train_hc']['ms_per_step'], '| lifter', b['train_lifter']['ms_per_step'], '| cpu', b['cpu_baseline']['value'])
PY
bash tools/prof_r6.sh > gpurun_out/prof_r6.log 2>&1
tail -4 gpurun_out/prof_r6.log
