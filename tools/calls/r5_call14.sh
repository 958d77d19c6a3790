# round 5 call 14: bisect the hipGraphLaunch crash over the tests of tests/test_gpu_autograd.py
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c14; mkdir -p $O
export TMPDIR=/tmp
for t in hrnet_coordinates hrnet_heatmap eval_routes lifter_loop lifter_two lifter_drop; do
  timeout 300 python tools/graph_crash_bisect.py $t > $O/$t.txt 2>&1; echo "$t: rc=$? $(grep -E 'graph case ok|Fatal|Error' $O/$t.txt | tail -1 | cut -c1-120)"
done
