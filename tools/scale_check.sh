#!/usr/bin/env bash
# Scaling check for a multi-GPU MI355X node (the builder's boxes have ONE GPU: this is the script the
# driver's 8-GPU box -- or anyone with more than one device -- runs).
#
#   bash tools/scale_check.sh [max_gpus]          (default: every visible device, powers of two)
#
# For N = 1, 2, 4, 8 ... it runs `python bench.py --gpus N` (bench.py spawns the N ranks itself: one process per
# GPU, backend nccl = RCCL over xGMI) and prints, per N:
#   * inference:  whole-job crops/s and the weak-scaling efficiency  value(N) / (N * value(1))
#                 (replicas, crops sharded by rank, NO collective on the data path -- expectation: ~1.0);
#   * train_hc:   whole-job crops/s, efficiency, ms/step, and the EXPOSED gradient-exchange time per step =
#                 ms/step(N) - ms/step(1): the flat 256 MB gradient is all-reduced in 32 MB slices on a
#                 high-priority communication stream while the backward still runs (egonet_amd/parallel.py);
#                 what is not hidden under the backward shows up here.  With EGONET_AMD_GRAD_OVERLAP=0 (one
#                 exchange after the backward) the same column is the full all-reduce time -- run both to see
#                 what the overlap buys:   EGONET_AMD_GRAD_OVERLAP=0 bash tools/scale_check.sh
# RCCL settings this path is written for (none are set here): defaults for NCCL_MIN/MAX_NCHANNELS (rings over
# the 7 xGMI links of a fully connected 8-GPU node), HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC; exported below).
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
MAX=${1:-$NDEV}
if [ "$NDEV" -lt 1 ]; then echo "no GPU visible"; exit 2; fi
OUT=${SCALE_OUT:-gpurun_out/scale_check}
mkdir -p "$OUT"
N=1
while [ "$N" -le "$MAX" ] && [ "$N" -le "$NDEV" ]; do
  echo "== bench.py --gpus $N" >&2
  python bench.py --gpus "$N" --steps "${STEPS:-50}" --warmup "${WARMUP:-5}" --no-cpu-baseline \
      > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err" || { echo "bench.py --gpus $N failed (see $OUT/bench_n$N.err)"; tail -5 "$OUT/bench_n$N.err"; exit 1; }
  N=$((N * 2))
done
python - "$OUT" <<'PY'
import glob, json, os, re, sys
out = sys.argv[1]
runs = {}
for f in glob.glob(os.path.join(out, 'bench_n*.json')):
    n = int(re.search(r'bench_n(\d+)\.json', f).group(1))
    lines = [ln for ln in open(f) if ln.startswith('{')]
    if lines:
        runs[n] = json.loads(lines[-1])
if 1 not in runs:
    sys.exit('no N = 1 run')
b = runs[1]
print('%5s | %14s %6s | %14s %6s %9s %12s' % ('N', 'infer crops/s', 'eff', 'train crops/s', 'eff', 'ms/step', 'exposed ms'))
for n in sorted(runs):
    r = runs[n]
    assert r['n_gpus'] == n, (n, r['n_gpus'])
    t, t1 = r.get('train_hc') or {}, b.get('train_hc') or {}
    line = '%5d | %14.0f %6.3f |' % (n, r['value'], r['value'] / (n * b['value']))
    if 'value' in t and 'value' in t1:
        line += ' %14.0f %6.3f %9.2f %12.2f' % (t['value'], t['value'] / (n * t1['value']), t['ms_per_step'],
                                                 t['ms_per_step'] - t1['ms_per_step'])
    print(line)
PY
