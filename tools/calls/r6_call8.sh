# round 6 call 8: bench with cfg 86 in the table; probe-library timing of cfg 88 / 90 vs 70 for the record
python bench.py --no-train --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/r6_bench_a.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_a.json').read())
print('bench', d['value'], d['unit'], d['ms_per_step'], 'roofline', d.get('roofline'))
PY
export EGONET_AMD_LIB=$PWD/tools/_build/libegonet_hip_probes.so
python tools/wino_probe.py --shape 64,64,64,48,48 --direct 0 --wino 70,80,88,90 --iters 20 2>&1 | grep " us "
