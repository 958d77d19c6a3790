"""The graphed small-batch forward (engine.HRNetEngine._forward_graphed) as a stand-alone case, run in a process of its
own by tests/test_gpu_models.py::test_small_batches_replay_their_program_as_a_hipgraph:

    EGONET_AMD_GRAPH_MAX_N=16 python tests/graph_case.py heatmap|coordinates
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from egonet_amd import configs, synth                                   # noqa: E402
from egonet_amd.model.heatmapModel import hrnet as hip_hrnet             # noqa: E402


def _model(cfg, seed):
    net = hip_hrnet.get_pose_net(cfg, is_train=False)
    sd = synth.synth_state_dict(net.state_dict(), seed=seed)
    net.load_state_dict(sd)
    return net.eval().cuda(), sd


def case(head):
    """[round 5] engine.HRNetEngine._forward_graphed: batches of <= EGONET_AMD_GRAPH_MAX_N crops (configs[4]'s 16-crop
    shard, configs[0]'s single crop: launch-bound) run their program eagerly twice, then replay it as ONE hipGraph with
    static input / output tensors.  Same bits as the eager engine for every call and every new input, fresh output
    tensors per call, the launch counter advances by the program's launches (no fallback), a weight change drops the
    graph with the program, and the default stream (which cannot be captured) is a legal caller."""
    from egonet_amd import _lib
    cfg = configs.tiny_config(head)
    net, sd = _model(cfg, 4)
    xs = [synth.synth_crops(3, 3, 64, 64, seed=20 + i).cuda() for i in range(5)]
    os.environ['EGONET_AMD_GRAPH_MAX_N'] = '0'
    net._engine = None
    want = []
    for x in xs:
        o = net._hip_engine().forward(x, decode_mode=1)
        want.append(o)
    assert not hasattr(net._hip_engine().program(xs[0], 1), 'static')
    os.environ['EGONET_AMD_GRAPH_MAX_N'] = '16'
    net._engine = None
    eng = net._hip_engine()
    L = _lib.lib()
    prev = None
    for i, x in enumerate(xs):
        n0 = L.egn_launch_count()
        got = eng.forward(x, decode_mode=1)              # on the default stream
        prog = eng.program(x, 1)
        assert prog.captured == (i >= 1), (i, prog.captured)
        nk = sum(1 for m in prog.meta if m['kind'] not in ('fork', 'join'))
        assert L.egn_launch_count() - n0 == nk
        flat_g = torch.utils._pytree.tree_leaves(got)
        flat_w = torch.utils._pytree.tree_leaves(want[i])
        assert len(flat_g) == len(flat_w)
        for a, b in zip(flat_g, flat_w):
            assert torch.equal(a, b)
        if prev is not None:                              # fresh tensors: the previous call's results are untouched
            for a, b in zip(prev, torch.utils._pytree.tree_leaves(want[i - 1])):
                assert torch.equal(a, b)
        prev = flat_g
    # module forward (the drop-in entry point) takes the same path; a weight change rebuilds program and graph
    with torch.no_grad():
        y1 = net(xs[0])
        first = next(net.parameters())
        first.mul_(1.25)
        y2 = net(xs[0])
        y3 = net(xs[0])
        y4 = net(xs[0])
    l1, l2, l3, l4 = (torch.utils._pytree.tree_leaves(t)[0] for t in (y1, y2, y3, y4))
    assert float((l1 - l2).abs().max()) > 0 and torch.equal(l2, l3) and torch.equal(l3, l4)


if __name__ == '__main__':
    case(sys.argv[1] if len(sys.argv) > 1 else 'heatmap')
    torch.cuda.synchronize()
    print('graph case ok')
