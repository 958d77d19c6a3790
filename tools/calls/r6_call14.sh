# round 6 call 14: same-box A/B of cfg 86 inside the network (bench line + per-class table), alternating
for rep in 1 2; do
  for skip in 86 ""; do
    EGONET_AMD_SKIP_CFG=$skip python bench.py --no-train --no-cpu-baseline --steps 30 --warmup 5 --profile-json gpurun_out/ab_$rep"_skip"$skip.json 2>/dev/null | tail -1 > gpurun_out/ab_line.json
    python - "$rep" "$skip" <<'PY'
import json, sys
b = json.loads(open('gpurun_out/ab_line.json').read())
d = json.load(open('gpurun_out/ab_%s_skip%s.json' % (sys.argv[1], sys.argv[2])))
c = {x['name']: x for x in d['classes']}
print('rep %s skip "%s": %.1f crops/s %.3f ms | 96->96@32x32 %.1f us | 48->48@64x64 %.1f us | 192 %.1f us | kernel sum %.3f ms' % (
    sys.argv[1], sys.argv[2], b['value'], b['ms_per_step'], c['conv3x3s1 96->96@32x32']['avg_us'], c['conv3x3s1 48->48@64x64']['avg_us'],
    c['conv3x3s1 192->192@16x16']['avg_us'], sum(x['ms'] for x in d['classes'])))
PY
  done
done
