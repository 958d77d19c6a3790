R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c6
mkdir -p $O
cd $R
timeout 900 python bench.py --no-train --no-cpu-baseline --steps 30 --profile-json $O/profile.json > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json,os,collections
o=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4c6/'
d=json.load(open(o+'profile.json'))
tot=sum(c['ms'] for c in d['classes'])
print('total %.3f ms over %d classes'%(tot,len(d['classes'])))
for c in d['classes']:
    print('%-36s %3d %7.1f us %6.3f ms %5.1f%%'%(c['name'],c['launches'],c['avg_us'],c['ms'],100*c['ms']/tot))
PY
