# round 5 call 18: which part of the pytest session matters for the hipGraphLaunch crash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c18; mkdir -p $O
export TMPDIR=/tmp
A=tests/test_gpu_autograd.py; G=tests/_graph_inproc.py
run() { n=$1; shift; timeout 400 python -m pytest "$@" -q -m gpu > $O/$n.txt 2>&1; echo "$n [$*]: rc=$? $(grep -E 'passed|failed|Fatal' $O/$n.txt | tail -1 | cut -c1-90)"; }
run base $A $G
run nofault -p no:faulthandler $A $G
run nocapture -s $A $G
run nowarn -p no:warnings $A $G
run hrnet_only "$A::test_reference_training_loop_on_the_native_tape_hrnet" $G
run lifter_only "$A::test_reference_training_loop_on_the_native_tape_lifter" "$A::test_lifter_bridge_with_dropout_and_two_forwards_before_backward" "$A::test_lifter_bridge_releases_a_forward_whose_graph_is_dropped" $G
run eval_only "$A::test_eval_mode_routes_and_escape_hatches" $G
