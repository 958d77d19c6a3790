R=$GRAFT_REPO_ROOT
cd $R
run() { echo -n "$1: "; env $2 timeout 300 python tools/train_hc_bench.py --steps 8 --warmup 3 $3 2>gpurun_out/r4c11_err.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print(d['ms_per_step'], d['loss'])
"; grep -i "priority range\|Error" gpurun_out/r4c11_err.txt | head -2; }
mkdir -p gpurun_out
run "default (two streams, equal priority)" "A=1" ""
run "one stream" "EGONET_AMD_WGRAD_STREAM=0" ""
run "main high (-1), wgrad normal" "A=1" "--main-priority -1"
run "main normal, wgrad low (+1)" "EGONET_AMD_WGRAD_PRIORITY=1" ""
run "main high (-1), wgrad low (+1)" "EGONET_AMD_WGRAD_PRIORITY=1" "--main-priority -1"
run "default again" "A=1" ""
