# round 5 call 19: the crash needs EARLIER graph captures in the process (EGONET_AMD_GRAPH_MAX_N=16 for the whole session): which test?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c19; mkdir -p $O
export TMPDIR=/tmp
export EGONET_AMD_GRAPH_MAX_N=16
A=tests/test_gpu_autograd.py; G=tests/_graph_inproc.py
run() { n=$1; shift; timeout 400 python -X faulthandler -m pytest "$@" -q -m gpu > $O/$n.txt 2>&1; echo "$n: rc=$? $(grep -E 'passed|failed|Fatal' $O/$n.txt | tail -1 | cut -c1-90)"; }
run all16 $A $G
run hrnet16 "$A::test_reference_training_loop_on_the_native_tape_hrnet" $G
run prof16 "$A::test_training_loop_runs_no_foreign_conv_kernels" $G
run eval16 "$A::test_eval_mode_routes_and_escape_hatches" $G
run lifter16 "$A::test_reference_training_loop_on_the_native_tape_lifter" "$A::test_lifter_bridge_with_dropout_and_two_forwards_before_backward" "$A::test_lifter_bridge_releases_a_forward_whose_graph_is_dropped" $G
