#!/usr/bin/env python
"""Which test of tests/test_gpu_autograd.py makes a LATER hipGraphLaunch crash the interpreter (profiles/
r5_graph_replay_crash.txt)?  Runs the named tests in this process (with the module's EGONET_AMD_AUTOTUNE=0 fixture), then
the graphed small-batch case (tests/graph_case.py).

    python tools/graph_crash_bisect.py hrnet_coordinates hrnet_heatmap eval_routes lifter_loop lifter_two lifter_drop
"""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

import test_gpu_autograd as T  # noqa: E402
import graph_case  # noqa: E402

CASES = {
    'hrnet_coordinates': lambda: T.test_reference_training_loop_on_the_native_tape_hrnet('coordinates'),
    'hrnet_heatmap': lambda: T.test_reference_training_loop_on_the_native_tape_hrnet('heatmap'),
    'profiler': T.test_training_loop_runs_no_foreign_conv_kernels,
    'eval_routes': T.test_eval_mode_routes_and_escape_hatches,
    'lifter_loop': T.test_reference_training_loop_on_the_native_tape_lifter,
    'lifter_two': T.test_lifter_bridge_with_dropout_and_two_forwards_before_backward,
    'lifter_drop': T.test_lifter_bridge_releases_a_forward_whose_graph_is_dropped,
}

if __name__ == '__main__':
    for name in sys.argv[1:]:
        os.environ['EGONET_AMD_AUTOTUNE'] = '0'
        CASES[name]()
        os.environ.pop('EGONET_AMD_AUTOTUNE', None)
        torch.cuda.synchronize()
        print('ran', name, flush=True)
    os.environ['EGONET_AMD_GRAPH_MAX_N'] = '16'
    graph_case.case('heatmap')
    torch.cuda.synchronize()
    print('graph case ok after', sys.argv[1:], flush=True)
