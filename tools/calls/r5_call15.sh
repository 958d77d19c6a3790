# round 5 call 15: the hipGraphLaunch crash: all tests of the autograd module in one process, then halves
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c15; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 400 python tools/graph_crash_bisect.py "$@" > $O/$n.txt 2>&1; echo "$n [$*]: rc=$? $(grep -E 'graph case ok|Fatal' $O/$n.txt | tail -1 | cut -c1-100)"; }
run all hrnet_coordinates hrnet_heatmap profiler eval_routes lifter_loop lifter_two lifter_drop
run noprof hrnet_coordinates hrnet_heatmap eval_routes lifter_loop lifter_two lifter_drop
run hrnet2 hrnet_coordinates hrnet_heatmap
run hrnet2_eval hrnet_coordinates hrnet_heatmap eval_routes
run lifters lifter_loop lifter_two lifter_drop
run eval_lifters eval_routes lifter_loop lifter_two lifter_drop
