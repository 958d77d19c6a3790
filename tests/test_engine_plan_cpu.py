"""Host logic of the engine without a GPU: the recorded launch program of an
HRNet (op list, launch lanes, fork/join regions) and the activation-arena
packing.  Two tensors may share arena bytes only if every use of one
happens-before the definition of the other -- also across concurrent lanes."""
import itertools

import pytest
import torch

from egonet_amd import configs, engine
from egonet_amd.model.heatmapModel import hrnet
from egonet_amd.model import FCmodel


def _record(cfg, n=2, lanes=True, chain=True):
    net = hrnet.get_pose_net(cfg, is_train=False).eval()
    eng = engine.HRNetEngine(net)
    eng.lanes = lanes
    eng.chain_regions = chain
    iw, ih = cfg['heatmapModel']['input_size']
    rec, nslots, shapes = eng._record(n, 3, ih, iw, None)
    return rec, nslots, shapes


@pytest.mark.parametrize('head', ['coordinates', 'heatmap'])
@pytest.mark.parametrize('lanes', [True, False])
@pytest.mark.parametrize('model', ['tiny', 'w48'])
def test_arena_never_aliases_live_tensors(model, head, lanes):
    # (the W48 topology -- 4 blocks per branch, several modules per stage -- is
    # what exposes bytes freed by one lane being handed to another lane)
    cfg = configs.tiny_config(head) if model == 'tiny' else configs.w48_config(head)
    rec, nslots, shapes = _record(cfg, lanes=lanes)
    total = rec.plan_arena()
    bufs = [b for b in rec.bufs if b.slot == engine.SLOT_ARENA and b.first >= 0]
    assert total > 0 and all(b.off + b.nbytes <= total for b in bufs)
    for a, b in itertools.combinations(bufs, 2):
        if a.off < b.off + b.nbytes and b.off < a.off + a.nbytes:      # byte ranges overlap
            a_then_b = all(rec.happens_before(u, b.first) for u in a.uses)
            b_then_a = all(rec.happens_before(u, a.first) for u in b.uses)
            assert a_then_b or b_then_a, (a.name, b.name)
    # packing really re-uses memory
    assert total < 0.5 * sum(b.nbytes for b in bufs)


def test_program_structure_w48():
    """HRNet-W48, coordinates head: one fused launch per conv (+ layout/fuse/ramps
    helpers) instead of ~1050 unfused ATen kernels."""
    rec, nslots, shapes = _record(configs.w48_config('coordinates'), n=1)
    kinds = [k for k, _ in rec.ops]
    convs = [op for k, op in rec.ops if k == 'conv']
    # 306 Conv2d modules: every one is exactly one launch -- except layer1's 1x1 convolutions around the 256-channel
    # tensor [round 5]: 4 x conv3 + 3 x conv1 + the downsample conv = 8 modules in 5 launches of csrc/conv_pw.hip
    pws = [op for k, op in rec.ops if k == 'pwpair']
    assert len(pws) == 5 and sum(2 if op['hn'] is not None else 1 for op in pws) == 8
    assert len(convs) + 8 == 306
    assert kinds.count('fuse') == 2 + 4 * 3 + 2 * 4 + 1      # one per fuse output of the 8 HR modules
    # branches + fuse outputs of 8 modules = 16 regions; the fuse region of a module runs on into the
    # branches of the next module of its stage (3 times in stage 3, twice in stage 4): 11
    assert kinds.count('fork') == kinds.count('join') == 11
    rec16, _, _ = _record(configs.w48_config('coordinates'), n=1, chain=False)
    kinds16 = [k for k, _ in rec16.ops]
    assert kinds16.count('fork') == kinds16.count('join') == 16
    assert [k for k in kinds16 if k not in ('fork', 'join')] == [k for k in kinds if k not in ('fork', 'join')]
    assert shapes['maps'] == (1, 33, 64, 64) and shapes['coords'] == (1, 33, 2)
    flops = sum(2.0 * op['ho'] * op['wo'] * op['cout'] * op['cin'] * op['kh'] * op['kw'] for op in convs)
    flops += sum(2.0 * op['m'] * 64 * 256 * (2 if op['hn'] is not None else 1) for op in pws)
    assert abs(flops / 1e9 - 42.035) < 0.02   # GFLOP per crop
    # every lane index stays inside the 4 launch lanes and lanes > 0 only occur inside regions
    inside = False
    for k, op in rec.ops:
        if k == 'fork':
            inside = True
        elif k == 'join':
            inside = False
        else:
            assert 0 <= op['lane'] < 4 and (inside or op['lane'] == 0)


def test_lifter_program_and_weight_blob():
    cfg = configs.w48_config()
    net = FCmodel.get_fc_model(1, cfg, 66, 96).eval()
    eng = engine.LifterEngine(net)
    rec = eng._record(7)
    convs = [op for k, op in rec.ops if k == 'conv']
    assert [(c['cin'], c['cout']) for c in convs] == [(66, 1024)] + [(1024, 1024)] * 4 + [(1024, 96)]
    assert convs[2]['act'] == (engine.ACT_RELU | engine.ACT_RES_AFTER) and convs[2]['res'] is not None
    blob = rec.weights_blob('cpu')
    # w1's packed weight: [nchunk=5][1][4][1024][4], chunk 0 / quad 0 / co 3 = w1.weight[3, 0:4]
    o = convs[0]['w'].off // 4
    w = blob[o:o + 5 * 4 * 1024 * 4].view(5, 1, 4, 1024, 4)
    assert torch.equal(w[0, 0, 0, 3], net.w1.weight[3, 0:4].detach())
    assert float(w[4, 0, 0, :, 2:].abs().sum()) == 0.0      # input channels 66, 67 are padding


def test_max_batch_and_chunk_sizes_of_oversized_batches(monkeypatch):
    """[round 6] HRNetEngine.max_batch / _forward_chunked (host logic, no GPU): the widest tensor of the network decides how
    many crops one program takes (32-bit byte offsets in the kernels: 2 GiB per tensor); larger batches are cut greedily
    into the tile table's batch sizes.  The reference accepts any loader batch (libs/trainer/trainer.py:113-125)."""
    net = hrnet.get_pose_net(configs.tiny_config('heatmap'), is_train=False).eval()
    eng = engine.HRNetEngine(net)
    monkeypatch.delenv('EGONET_AMD_MAX_TENSOR_BYTES', raising=False)
    assert eng.max_batch(3, 256, 256) == 511                      # layer1's 256 channels at 64 x 64: 4 MiB per crop
    assert eng.max_batch(3, 128, 128) == 2047 and eng.max_batch(3, 512, 512) == 127
    assert eng.max_batch(3, 2048, 2048) == 7                      # (never below one crop per program: max(1, ...))
    monkeypatch.setenv('EGONET_AMD_MAX_TENSOR_BYTES', str(4 * 256 * 16 * 16 * 5))
    assert eng.max_batch(3, 64, 64) == 5
    seen = []

    def fake_forward(x, decode_mode=None, timed=False, slot=0):   # records the chunks instead of running them
        seen.append(x.shape[0])
        n = x.shape[0]
        maps = torch.arange(n, dtype=torch.float32).view(n, 1, 1, 1) + 100 * len(seen)
        if decode_mode is None:
            return maps
        return (maps, (maps.view(n, 1, 1).expand(n, 1, 2), maps.view(n, 1, 1), maps.view(n, 1).int()))
    monkeypatch.setattr(eng, 'forward', fake_forward)
    out, (xy, mx, idx) = eng._forward_chunked(torch.zeros(13, 3, 64, 64), 1, False, 0, 5)
    assert seen == [4, 4, 4, 1] and out.shape[0] == 13 and xy.shape == (13, 1, 2) and idx.shape == (13, 1)
    assert out.view(-1).tolist() == [100., 101., 102., 103., 200., 201., 202., 203., 300., 301., 302., 303., 400.]
    seen.clear()
    eng._forward_chunked(torch.zeros(520, 3, 8, 8), None, False, 0, 511)
    assert seen == [128, 128, 128, 128, 8]
