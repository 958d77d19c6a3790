"""Drop-in boundary on CPU: checkpoint layout, factory behaviour and the
torch-functional (CPU tensor) forward of the module mirrors vs the oracle."""
import zlib

import numpy as np
import pytest
import torch

from conftest import golden
from egonet_amd import configs, synth
from egonet_amd.model.heatmapModel import hrnet
from egonet_amd.model import FCmodel
from oracle import hrnet_oracle, lifter_oracle


def _key_crc(sd):
    return zlib.crc32('\n'.join('%s %s' % (k, tuple(v.shape)) for k, v in sd.items()).encode())


@pytest.mark.parametrize('head', ['coordinates', 'heatmap'])
def test_checkpoint_layout_equals_reference(head):
    g = golden('hrnet_w48_outputs.npz')
    net = hrnet.get_pose_net(configs.w48_config(head), is_train=False)
    sd = net.state_dict()
    assert len(sd) == int(g[head + '/n_keys']) == (1828 if head == 'coordinates' else 1754)
    assert sum(p.numel() for p in net.parameters()) == int(g[head + '/n_params'])
    assert _key_crc(sd) == int(g[head + '/key_crc'])        # names, shapes AND order
    assert 'coor_maps' not in sd and not any('coor_maps' in k for k in sd)


def test_cpu_forward_matches_oracle_and_factory_side_effects(capsys):
    cfg = configs.tiny_config('coordinates')
    cfg['heatmapModel']['extra']['freeze_layers'] = ['conv1', 'layer1']
    net = hrnet.get_pose_net(cfg, is_train=True)           # init_weights path (N(0,1e-3) convs, BN 1/0)
    assert 'conv1.weight freezed during training.' in capsys.readouterr().out
    assert not net.conv1.weight.requires_grad and net.conv2.weight.requires_grad
    assert abs(float(net.conv2.weight.std()) - 1e-3) < 3e-4 and float(net.bn2.weight.min()) == 1.0
    sd = synth.synth_state_dict(net.state_dict(), seed=3)
    net.load_state_dict(sd)
    net.eval()
    x = synth.synth_crops(2, 3, 64, 64, seed=5)
    with torch.no_grad():
        maps, coords = net(x)
    om, oc = hrnet_oracle.hrnet_forward(sd, cfg, x)
    assert torch.equal(maps, om) and torch.equal(coords, oc)
    with pytest.raises(ValueError):
        c2 = configs.tiny_config()
        c2['heatmapModel']['pretrained'] = '/nonexistent/HC.pth'
        hrnet.get_pose_net(c2, is_train=True)
    with pytest.raises(NotImplementedError):
        c3 = configs.tiny_config()
        c3['heatmapModel']['head_type'] = 'bogus'
        hrnet.get_pose_net(c3, is_train=False)
    # add_xy widens conv1 to 5 input channels keeping the RGB filters
    c4 = configs.tiny_config()
    c4['heatmapModel']['add_xy'] = True
    assert hrnet.get_pose_net(c4, is_train=False).conv1.weight.shape == (64, 5, 3, 3)


def test_lifter_mirror_cpu_and_training_mode():
    cfg = configs.w48_config()
    net = FCmodel.get_fc_model(1, cfg, 66, 96)
    sd = synth.synth_state_dict(net.state_dict(), seed=2)
    net.load_state_dict(sd)
    assert list(sd)[:4] == ['w1.weight', 'w1.bias', 'batch_norm1.weight', 'batch_norm1.bias']
    net.eval()
    x = torch.randn(5, 66, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        assert torch.allclose(net(x), lifter_oracle.lifter_forward(sd, x), atol=1e-6)
    net.train()                                            # dropout + batch statistics: autograd path works
    y = net(x)
    y.sum().backward()
    assert net.w1.weight.grad is not None
    assert len(FCmodel.get_cascade()) == 0


def test_pixel_shuffle_and_angle_heads_match_the_reference_on_cpu():
    """The remaining head variants (reference hrnet.py:373-422, 598-611): same state_dict keys in the
    same order, and the module's CPU forward reproduces the reference's outputs
    (tests/golden/hrnet_tiny_pixshuf.npz, hrnet_tiny_angle.npz)."""
    import json
    import numpy as np
    from conftest import golden, fixture_cfg, sd_crc, require_same_rng
    from egonet_amd import synth
    from egonet_amd.model.heatmapModel import hrnet
    for tag, shape in (('tiny_pixshuf', (2, 5, 32, 32)), ('tiny_angle', (3, 2))):
        g = golden('hrnet_%s.npz' % tag)
        cfg = fixture_cfg(g)
        net = hrnet.get_pose_net(cfg, is_train=False).eval()
        assert list(net.state_dict()) == json.loads(str(g['keys']))
        sd = synth.synth_state_dict(net.state_dict(), seed=3)
        require_same_rng(sd_crc(sd), g['sd_crc'], 'weights')
        net.load_state_dict(sd)
        iw, ih = cfg['heatmapModel']['input_size']
        x = synth.synth_crops(int(g['n']), 3, ih, iw, seed=5)
        with torch.no_grad():
            out = net(x)
        assert tuple(out.shape) == shape == g['out'].shape
        np.testing.assert_allclose(out.numpy(), g['out'], rtol=0, atol=1e-5)


def test_forward_hooks_fire_like_get_model_summary_expects():
    """libs/common/utils.py:91-95 (get_model_summary) registers forward hooks on the leaf modules and
    runs one CPU forward (tools/train_IGRs.py:54-57): every conv / BatchNorm leaf is called once."""
    from egonet_amd import configs
    from egonet_amd.model.heatmapModel import hrnet
    net = hrnet.get_pose_net(configs.tiny_config('coordinates'), is_train=False).eval()
    leaves = [m for m in net.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.BatchNorm2d))]
    calls = []
    hooks = [m.register_forward_hook(lambda mod, i, o: calls.append(id(mod))) for m in leaves]
    net(torch.randn(1, 3, 64, 64))
    for h in hooks:
        h.remove()
    assert sorted(calls) == sorted(id(m) for m in leaves)


def test_autograd_bridge_opt_outs_are_explicit():
    """the train-mode bridge replaces the module graph by ONE native autograd node, so anything that
    needs the graph must take the torch forward: submodule hooks (they would not fire) and nn.DataParallel replicas
    (re-created every forward: the bridge would be rebuilt per step).  Checked on the predicate (no GPU needed)."""
    import torch
    from egonet_amd import configs
    from egonet_amd.model.heatmapModel import hrnet as hip_hrnet
    from egonet_amd.model import FCmodel as hip_fc
    net = hip_hrnet.get_pose_net(configs.tiny_config('coordinates'), is_train=True)
    assert net._native_autograd_ok()
    h = net.stage2[0].branches[0][0].conv1.register_forward_hook(lambda m, i, o: None)
    assert not net._native_autograd_ok()
    h.remove()
    assert net._native_autograd_ok()
    h = net.layer1[0].register_full_backward_hook(lambda m, gi, go: None)
    assert not net._native_autograd_ok()
    h.remove()
    h = net.register_forward_hook(lambda m, i, o: None)          # a hook on the module ITSELF fires either way
    assert net._native_autograd_ok()
    h.remove()
    net._is_replica = True                                        # what torch.nn.parallel.replicate sets
    assert not net._native_autograd_ok()
    del net._is_replica

    class _X(object):                                             # the lifter's predicate looks at the input too
        is_cuda, requires_grad, shape = True, False, (8, 10)
    lif = hip_fc.get_fc_model(1, configs.tiny_config(), 10, 12).train()
    assert lif._native_autograd_ok(_X())
    h = lif.res_blocks[0].w1.register_forward_pre_hook(lambda m, i: None)
    assert not lif._native_autograd_ok(_X())
    h.remove()
    assert lif._native_autograd_ok(_X())


def test_dropout_seed_is_mixed_with_the_rank(monkeypatch):
    """Data-parallel ranks seeded alike must not draw identical keep masks (the reference's DataParallel replicas use
    per-device generators): seed ^ rank * golden ratio, rank 0 / single process unchanged."""
    import torch.distributed as dist
    from egonet_amd import train_lifter
    s = 0x1234567890ABCDEF & ((1 << 62) - 1)
    assert train_lifter._rank_mixed_seed(s) == s
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    seeds = []
    for r in range(4):
        monkeypatch.setattr(dist, 'get_rank', lambda r=r: r)
        seeds.append(train_lifter._rank_mixed_seed(s))
    assert seeds[0] == s and len(set(seeds)) == 4 and all(0 <= v < (1 << 62) for v in seeds)


def test_hook_cache_follows_the_module_tree_and_stays_out_of_pickles():
    """[round 6, ADVICE r5] heatmapModel.hrnet._has_submodule_hooks caches the submodules' hook dictionaries; the cache is
    keyed on a generation counter that every submodule registration bumps, so a block that is replaced AFTER the cache
    was filled and then given a hook is seen (the native path would skip the hook silently otherwise); and
    ``torch.save(model)`` carries neither the cache nor per-process device state."""
    import copy
    import io
    from egonet_amd.model.heatmapModel import hrnet as H
    net = H.get_pose_net(configs.tiny_config('heatmap'), is_train=False)
    assert not H._has_submodule_hooks(net) and '_hook_dicts' in net.__dict__
    net.stage2[0].branches[0][0] = copy.deepcopy(net.stage2[0].branches[0][0])       # a new object, unknown to the cache
    handle = net.stage2[0].branches[0][0].register_forward_hook(_noop_hook)
    assert H._has_submodule_hooks(net)
    handle.remove()
    assert not H._has_submodule_hooks(net)
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert '_hook_dicts' not in back.__dict__ and '_bridge' not in back.__dict__ and back._engine is None
    assert list(back.state_dict().keys()) == list(net.state_dict().keys())


def _noop_hook(module, inputs, output):
    return None


def test_apply_dropout_state_is_seen_by_the_lifter():
    """[round 6] libs/trainer/trainer.py:424-428 (testing_settings.apply_dropout): ``model.eval()`` then ``Dropout.train()``.
    FCModel._dropout_active reports that state (the eval-mode HIP program has no dropout: such a forward must take the
    module's torch graph); plain eval / train states do not trigger it, p = 0 never does."""
    from egonet_amd.model import FCmodel
    cfg = configs.tiny_config()
    net = FCmodel.get_fc_model(1, cfg, 10, 12).eval()
    assert not net._dropout_active()

    def apply_dropout(m):
        if type(m) == torch.nn.Dropout:
            m.train()
    net.apply(apply_dropout)
    assert net._dropout_active() == (cfg['FCModel']['dropout'] > 0)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.5
    assert net._dropout_active()
    a, b = net(torch.ones(4, 10)), net(torch.ones(4, 10))
    assert not torch.equal(a, b)                 # masks are drawn (CPU tensors run the torch graph anyway)
    net.eval()
    assert not net._dropout_active()
    assert torch.equal(net(torch.ones(4, 10)), net(torch.ones(4, 10)))
