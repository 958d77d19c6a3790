"""Native HRNet training step (egonet_amd.train_hrnet, BASELINE config 4 per GPU)
on MI355X against (a) two iterations of the REFERENCE (tests/golden/hrnet_train.npz:
reference model, reference JointsCompositeLoss, torch Adam) and (b) the CPU
training oracle at the full W48 / 256x256 size.

Tolerances: fp32 throughout; gradients are compared relative to each tensor's
largest entry.  Adam's first steps move every entry by ~lr*sign(grad), so an
entry whose gradient is rounding noise can step the other way -- parameters
after the update are compared with an outlier budget, as in the CPU oracle test.
"""
import json

import numpy as np
import pytest
import torch

from conftest import golden, fixture_cfg, sd_crc, require_same_rng
from egonet_amd import configs, synth
from egonet_amd.model.heatmapModel import hrnet as hip_hrnet
from egonet_amd.train_hrnet import HRNetTrainStep
from oracle.hrnet_train_oracle import HRNetTrainOracle
from train_checks import LayerChecks, gradient_agreement

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_autotune(monkeypatch):
    monkeypatch.setenv('EGONET_AMD_AUTOTUNE', '0')      # cost-model tile choice: keeps the tests short


def _tiny_model(cfg, seed=21):
    net = hip_hrnet.get_pose_net(cfg, is_train=False)
    sd = synth.synth_state_dict(net.state_dict(), seed=seed)
    net.load_state_dict(sd)
    return net.cuda().train(), sd


def _rel_err(got, want):
    scale = float(np.abs(want).max())
    return float(np.abs(got - want).max()) / max(scale, 1e-12)


def test_first_step_gradients_vs_reference():
    g = golden('hrnet_train.npz')
    cfg = fixture_cfg(g)
    net, sd = _tiny_model(cfg)
    require_same_rng(sd_crc(sd), g['sd_crc'], 'weights')
    tr = HRNetTrainStep(net, lr=1e-3)
    x = synth.synth_crops(4, 3, 64, 64, seed=30).cuda()
    loss = tr.step(x, torch.from_numpy(g['target'][0]).cuda(), torch.from_numpy(g['joints'][0]), update=False)
    assert abs(float(loss.item()) - float(g['losses'][0])) < 2e-5 * abs(float(g['losses'][0]))
    np.testing.assert_allclose(tr.last_maps.cpu().numpy(), g['maps1'], rtol=0, atol=2e-4)
    # coordinates are normalised by the 64-pixel crop: the parity bar of 1e-3 px is 1.5e-5 here (which tile
    # configuration -- direct or Winograd -- the tuner measures fastest on this box moves them by ~1e-5)
    np.testing.assert_allclose(tr.last_coords.cpu().numpy(), g['coords1'], rtol=0, atol=1.5e-5)
    named = dict(net.named_parameters())
    order = json.loads(str(g['param_order']))
    assert list(named) == order
    # one ReLU tie resolved the other way moves the tensors below it by ~1e-2 relative
    # (tests/train_checks.py); without one the agreement is ~1e-5
    norms = np.array([float(named[k].grad.double().norm()) for k in order])
    np.testing.assert_allclose(norms, g['grad_norms'], rtol=3e-2, atol=1e-8)
    assert np.median(np.abs(norms / np.maximum(g['grad_norms'], 1e-30) - 1)) < 1e-3
    errs = [_rel_err(named[k].grad.cpu().numpy(), g['g1/' + k]) for k in json.loads(str(g['keys']))]
    assert max(errs) < 5e-2 and np.median(errs) < 2e-3, errs


@pytest.mark.parametrize('wino', [False, True])
def test_two_steps_vs_reference(wino, monkeypatch):
    """Two optimisation steps against the reference's fixture.  wino=False pins the direct kernels (tight
    bounds); wino=True (EGONET_AMD_WINO=1) puts every eligible layer (32-multiple widths) on the Winograd kernels:
    the first step agrees as tightly, after it Adam's scale-free update turns the sign of gradients that
    are numerically zero into +-lr moves, so the second loss is only bounded to 2e-3."""
    g = golden('hrnet_train.npz')
    cfg = fixture_cfg(g)
    net, _ = _tiny_model(cfg)
    if wino:
        monkeypatch.setenv('EGONET_AMD_WINO', '1')
    tr = HRNetTrainStep(net, lr=1e-3)
    tr.allow_wino = wino
    losses = []
    for it in range(2):
        x = synth.synth_crops(4, 3, 64, 64, seed=30 + it).cuda()
        loss = tr.step(x, torch.from_numpy(g['target'][it]).cuda(), torch.from_numpy(g['joints'][it]))
        losses.append(float(loss.item()))
    np.testing.assert_allclose(losses[0], g['losses'][0], rtol=2e-5)
    np.testing.assert_allclose(losses[1], g['losses'][1], rtol=2e-3 if wino else 2e-4)
    fin = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    # after two Adam steps every entry has moved by ~2e-3 (lr * m/sqrt(v) is scale free):
    # agreement to a few 1e-6 is the rule; an entry whose two gradients nearly cancel
    # amplifies a ReLU-tie difference (tests/train_checks.py) -- those are bounded in number
    ds = [np.abs(fin[k] - g['p2/' + k]).ravel() for k in json.loads(str(g['keys']))]
    if wino:        # over all sampled tensors together (small tensors scatter: a dozen flipped +-lr entries of 66)
        d = np.concatenate(ds)
        assert np.median(d) < 2e-4 and np.mean(d > 5e-4) < 0.1, (float(np.median(d)), float(np.mean(d > 5e-4)))
    else:
        for k, d in zip(json.loads(str(g['keys'])), ds):
            assert np.median(d) < 5e-6 and np.mean(d > 5e-4) < 0.02, (k, float(np.median(d)), float(np.mean(d > 5e-4)))
    for k in ('bn1.running_mean', 'bn1.running_var', 'stage3.0.branches.2.0.bn1.running_var',
              'head2.1.bn2.running_mean'):
        np.testing.assert_allclose(fin[k], g['p2/' + k], rtol=5e-3 if wino else 1e-3, atol=5e-4 if wino else 1e-5,
                                   err_msg=k)
    assert int(fin['bn1.num_batches_tracked']) == 2


def test_heatmap_head_vs_oracle():
    cfg = configs.tiny_config('heatmap')
    net, sd = _tiny_model(cfg, seed=5)
    gen = torch.Generator().manual_seed(1)
    x = synth.synth_crops(3, 3, 64, 64, seed=2)
    tgt = torch.rand(3, 5, 16, 16, generator=gen)
    orc = HRNetTrainOracle(sd, cfg, lr=1e-3, w_coor=0.0)
    want_loss, want_maps, _ = orc.step(x, tgt, None, update=False)
    tr = HRNetTrainStep(net, lr=1e-3, w_coor=0.0)
    loss = tr.step(x.cuda(), tgt.cuda(), None, update=False)
    assert abs(float(loss.item()) - want_loss) < 2e-5 * abs(want_loss)
    np.testing.assert_allclose(tr.last_maps.cpu().numpy(), want_maps.numpy(), rtol=0, atol=2e-4)
    # a ReLU tie resolved the other way changes every gradient below it by a finite
    # amount (tests/train_checks.py): the criterion tolerates a flipped gate or two
    gl2, cos, med = gradient_agreement(dict(net.named_parameters()), orc.grads())
    assert cos > 0.9999 and gl2 < 1e-2 and med < 5e-3, (gl2, cos, med)


def test_heatmap_head_with_target_weights_vs_oracle():
    """JointsMSELoss(use_target_weight=True) (libs/loss/function.py:22-46, tools/train_IGRs.py:42): both maps times
    target_weight[:, k] before the per-joint MSE -- invisible joints (weight 0) drop out of loss and gradient.  The
    oracle's restatement is pinned on the reference's class (tests/golden/jmse_loss.npz)."""
    cfg = configs.tiny_config('heatmap')
    net, sd = _tiny_model(cfg, seed=5)
    gen = torch.Generator().manual_seed(1)
    x = synth.synth_crops(3, 3, 64, 64, seed=2)
    tgt = torch.rand(3, 5, 16, 16, generator=gen)
    tw = (torch.rand(3, 5, 1, generator=gen) > 0.3).float() * (0.5 + torch.rand(3, 5, 1, generator=gen))
    assert float(tw.min()) == 0.0
    orc = HRNetTrainOracle(sd, cfg, lr=1e-3, w_coor=0.0)
    want_loss, want_maps, _ = orc.step(x, tgt, None, update=False, target_weight=tw)
    plain_loss, _, _ = HRNetTrainOracle(sd, cfg, lr=1e-3, w_coor=0.0).step(x, tgt, None, update=False)
    assert abs(want_loss - plain_loss) > 1e-3 * abs(plain_loss)
    tr = HRNetTrainStep(net, lr=1e-3, w_coor=0.0, use_target_weight=True)
    loss = tr.step(x.cuda(), tgt.cuda(), None, update=False, target_weight=tw.cuda())
    assert abs(float(loss.item()) - want_loss) < 2e-5 * abs(want_loss)
    np.testing.assert_allclose(tr.last_maps.cpu().numpy(), want_maps.numpy(), rtol=0, atol=2e-4)
    gl2, cos, med = gradient_agreement(dict(net.named_parameters()), orc.grads())
    assert cos > 0.9999 and gl2 < 1e-2 and med < 5e-3, (gl2, cos, med)
    with pytest.raises(ValueError):
        tr.step(x.cuda(), tgt.cuda(), None, update=False)            # the flag is on: the weights are required


@pytest.mark.parametrize('hm_type,coor_type,optim', [
    ('sl1', 'mse', dict(optim_type='sgd', momentum=0.9, weight_decay=1e-3)),
    ('l1', 'sl1', dict(optim_type='adam', weight_decay=1e-2)),
    ('mse', 'l1', dict(optim_type='sgd', momentum=0.0, weight_decay=0.0)),
])
def test_loss_criteria_and_optimizers_of_the_config_system_vs_oracle(hm_type, coor_type, optim):
    """loss_dict (function.py:17-20) for the heat-map and coordinate terms, torch.optim.SGD(momentum,
    weight_decay) / Adam(weight_decay) (optimizer.py:8-40): first-step loss and gradients, and the
    parameters after TWO updates (the second SGD step reads the momentum buffer), against the CPU oracle."""
    cfg = configs.tiny_config('coordinates')
    net, sd = _tiny_model(cfg, seed=11)
    gen = torch.Generator().manual_seed(4)
    xs = [synth.synth_crops(3, 3, 64, 64, seed=20 + i) for i in range(2)]
    tg = [torch.rand(3, 5, 16, 16, generator=gen) * 3 for _ in range(2)]       # |d| > 1 occurs: both sl1 branches
    jt = [torch.rand(3, 5, 2, generator=gen) * 64 for _ in range(2)]
    lr = 1e-2 if optim['optim_type'] == 'sgd' else 1e-3
    orc = HRNetTrainOracle(sd, cfg, lr=lr, cr=dict(hm_type=hm_type, coor_type=coor_type), optim=optim)
    tr = HRNetTrainStep(net, lr=lr, hm_type=hm_type, coor_type=coor_type, **optim)
    tr.allow_wino = False
    want, _, _ = orc.step(xs[0], tg[0], jt[0], update=False)
    loss = tr.step(xs[0].cuda(), tg[0].cuda(), jt[0], update=False)
    assert abs(float(loss.item()) - want) < 2e-5 * abs(want)
    gl2, cos, med = gradient_agreement(dict(net.named_parameters()), orc.grads())
    assert cos > 0.9999 and gl2 < 1e-2 and med < 5e-3, (gl2, cos, med)
    for i in range(2):
        orc.step(xs[i], tg[i], jt[i])
        tr.step(xs[i].cuda(), tg[i].cuda(), jt[i])
    named = dict(net.named_parameters())
    # the two updates themselves (a ReLU tie resolved the other way moves the gradients below it by ~1e-2
    # relative, tests/train_checks.py -- the bounds are those of the gradient checks above)
    un = np.concatenate([(named[k].detach().cpu() - sd[k]).numpy().ravel() for k in orc.param_keys]).astype(np.float64)
    uo = np.concatenate([(orc.sd[k].detach() - sd[k]).numpy().ravel() for k in orc.param_keys]).astype(np.float64)
    if optim['optim_type'] == 'sgd':           # linear in the gradients (+ momentum, weight decay)
        rel = np.linalg.norm(un - uo) / np.linalg.norm(uo)
        cosu = float(un @ uo / (np.linalg.norm(un) * np.linalg.norm(uo)))
        assert rel < 2e-2 and cosu > 0.9998, (rel, cosu)
    else:                                       # Adam: +-lr per step where a gradient is numerically zero
        d = np.abs(un - uo)
        assert np.median(d) < 5e-5 and d.max() <= 2.1 * 2 * lr and np.mean(d > lr) < 0.05, \
            (float(np.median(d)), float(d.max()), float(np.mean(d > lr)))


def test_step_with_winograd_forward_and_data_gradient_convs_vs_oracle(monkeypatch):
    """48/96/192/384-channel topology (the widths of W48, one block per branch, 64 x 64 input): with
    EGONET_AMD_WINO=1 every 3x3 stride-1 forward AND data-gradient convolution runs on the fused
    Winograd kernels (csrc/conv_wino.hip; filters transformed on the device by the step's batched
    pack).  Loss, maps and gradients against the CPU training oracle, and the second step (the
    batched pack of direct + Winograd filters in one launch) against the oracle's second step."""
    from egonet_amd import _lib
    monkeypatch.setenv('EGONET_AMD_WINO', '1')
    cfg = configs.tiny_config('coordinates', width=48)
    net, sd = _tiny_model(cfg, seed=7)
    gen = torch.Generator().manual_seed(2)
    x = synth.synth_crops(2, 3, 64, 64, seed=3)
    tgt = torch.rand(2, 5, 16, 16, generator=gen)
    jt = torch.rand(2, 5, 2, generator=gen) * 64
    orc = HRNetTrainOracle(sd, cfg, lr=1e-3)
    tr = HRNetTrainStep(net, lr=1e-3)
    for it in range(2):
        want_loss, want_maps, _ = orc.step(x, tgt, jt, update=True)
        loss = tr.step(x.cuda(), tgt.cuda(), jt, update=True)
        # (after one Adam step every entry has moved by ~lr * sign(g): entries whose gradient is rounding
        #  noise step either way, so the second iteration agrees to ~1e-4 like test_two_steps_vs_reference)
        assert abs(float(loss.item()) - want_loss) < (5e-5 if it == 0 else 1e-3) * abs(want_loss), it
        if it == 0:
            np.testing.assert_allclose(tr.last_maps.cpu().numpy(), want_maps.numpy(), rtol=0, atol=3e-4)
            gl2, cos, med = gradient_agreement(dict(net.named_parameters()), orc.grads())
            assert cos > 0.9999 and gl2 < 1e-2 and med < 5e-3, (gl2, cos, med)
    kinds = {code for (_, code) in tr.packs.entries}
    assert kinds == {0, 1, 2, 3}                 # direct + Winograd filters, forward + data gradient
    assert tr.packs.table is not None            # the second step packed them all in one launch
    n_wino = sum(1 for (_, code) in tr.packs.entries if code & 2)
    assert n_wino >= 2 * 16                      # 2 x (4 + 3 + 2 + ... ) 3x3 s1 convs of the branches / head
    # the one-launch pack of every filter == the single-filter entry points, bit for bit
    L = _lib.lib()
    st = _lib.current_stream()
    assert tr.packs.pack_all(st)
    for (_, code), (w, wp) in tr.packs.entries.items():
        one = torch.full_like(wp, float('nan'))
        cout, cin, kh, kw = w.shape
        if code & 2:
            _lib.check(L.egn_wino_pack_weight_f32(_lib.ptr(w), cout, cin, code & 1, _lib.ptr(one), st))
        else:
            _lib.check(L.egn_pack_conv_weight_f32(_lib.ptr(w), cout, cin, kh, kw, code & 1, _lib.ptr(one), st))
        assert torch.equal(one, wp), (tuple(w.shape), code)


def test_frozen_prefix_gets_no_gradient_and_no_update():
    cfg = configs.tiny_config('coordinates')
    net, sd = _tiny_model(cfg, seed=9)
    for name, p in net.named_parameters():
        if name.startswith(('conv1', 'bn1', 'layer1')):
            p.requires_grad = False
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    tr = HRNetTrainStep(net, lr=1e-3)
    gen = torch.Generator().manual_seed(3)
    x = synth.synth_crops(2, 3, 64, 64, seed=4).cuda()
    tr.step(x, torch.rand(2, 5, 16, 16, generator=gen).cuda(), torch.rand(2, 5, 2, generator=gen) * 64)
    after = net.state_dict()
    assert torch.equal(after['conv1.weight'], before['conv1.weight'])
    assert torch.equal(after['layer1.2.bn3.bias'], before['layer1.2.bn3.bias'])
    assert not torch.equal(after['stage2.0.branches.0.0.conv1.weight'], before['stage2.0.branches.0.0.conv1.weight'])
    assert not torch.equal(after['bn1.running_mean'], before['bn1.running_mean'])   # train-mode BN still tracks


def test_targets_drawn_on_the_device_give_the_same_step():
    """step(target=None) draws the Gaussian maps from the joints on the GPU
    (img_proc.py:347-409); same loss as with the oracle's host-side maps."""
    from oracle import targets_oracle
    cfg = configs.tiny_config('coordinates')
    gen = torch.Generator().manual_seed(6)
    x = synth.synth_crops(3, 3, 64, 64, seed=4).cuda()
    jt = torch.rand(3, 5, 2, generator=gen, dtype=torch.float64) * 80 - 8          # some joints off the crop
    vis = (torch.rand(3, 5, generator=gen) > 0.2).float()
    maps, w = targets_oracle.generate_target_batch(jt.numpy(), vis.numpy(), (64, 64), (16, 16), 1)
    losses = []
    for target in (torch.from_numpy(maps).cuda(), None):
        net, _ = _tiny_model(cfg, seed=9)
        tr = HRNetTrainStep(net, lr=1e-3, sigma=1)
        losses.append(float(tr.step(x, target, jt, joints_vis=vis).item()))
        if target is None:
            np.testing.assert_array_equal(tr.last_target_weight.cpu().numpy(), w)
    assert abs(losses[0] - losses[1]) < 1e-6 * abs(losses[0])


def test_graphed_hrnet_step_equals_eager_steps():
    """~1000 launches of a tiny HRNet iteration captured as one hipGraph: replays are
    bit-identical to eager steps (device-side Adam counter, static shapes)."""
    from egonet_amd.graph import GraphedStep
    cfg = configs.tiny_config('coordinates')
    gen = torch.Generator().manual_seed(2)
    xs = [synth.synth_crops(2, 3, 64, 64, seed=40 + i).cuda() for i in range(4)]
    tg = torch.rand(4, 2, 5, 16, 16, generator=gen).cuda()
    jt = (torch.rand(4, 2, 5, 2, generator=gen) * 64).cuda()
    out = []
    for graphed in (False, True):
        net, _ = _tiny_model(cfg, seed=9)
        tr = HRNetTrainStep(net, lr=1e-3)
        if graphed:
            before = {k: v.clone() for k, v in net.state_dict().items()}
            g = GraphedStep(tr, xs[0], tg[0], jt[0], warmup=1)       # the warm-up iteration is undone
            for k, v in net.state_dict().items():
                assert torch.equal(v, before[k]), k
            losses = [float(g(xs[i], tg[i], jt[i]).item()) for i in range(1, 4)]
        else:
            losses = [float(tr.step(xs[i], tg[i], jt[i]).item()) for i in range(1, 4)]
        out.append((net, losses, tr.flat.t))
    assert out[0][2] == out[1][2] == 3
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-12)       # the loss sum uses atomics: order varies
    for (k, a), (_, b) in zip(out[0][0].state_dict().items(), out[1][0].state_dict().items()):
        assert torch.equal(a, b), k


def test_inference_after_training_uses_the_updated_weights():
    cfg = configs.tiny_config('coordinates')
    net, _ = _tiny_model(cfg, seed=9)
    x = synth.synth_crops(2, 3, 64, 64, seed=4).cuda()
    net.eval()
    with torch.no_grad():
        m0, _ = net(x)
    net.train()
    tr = HRNetTrainStep(net, lr=1e-2)
    gen = torch.Generator().manual_seed(3)
    tr.step(x, torch.rand(2, 5, 16, 16, generator=gen).cuda(), torch.rand(2, 5, 2, generator=gen) * 64)
    net.eval()
    with torch.no_grad():
        m1, _ = net(x)
    assert float((m1 - m0).abs().max()) > 1e-4


def test_w48_gradients_vs_oracle_full_size():
    """Full HRNet-W48 at 256x256, 2 crops: every parameter gradient against the CPU
    oracle (torch autograd on the functional restatement)."""
    cfg = configs.w48_config('coordinates')
    net = hip_hrnet.get_pose_net(cfg, is_train=False)
    sd = synth.synth_state_dict(net.state_dict(), seed=1)
    net.load_state_dict(sd)
    gen = torch.Generator().manual_seed(12)
    x = synth.synth_crops(2, 3, 256, 256, seed=13)
    tgt = torch.rand(2, 33, 64, 64, generator=gen)
    jt = torch.rand(2, 33, 2, generator=gen) * 256
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    orc = HRNetTrainOracle(sd, cfg, lr=1e-3)
    want_loss, want_maps, want_coords = orc.step(x, tgt, jt, update=False)
    net = net.cuda().train()
    tr = HRNetTrainStep(net, lr=1e-3)
    with LayerChecks(tr) as chk:
        loss = tr.step(x.cuda(), tgt.cuda(), jt, update=False)
    assert abs(float(loss.item()) - want_loss) < 5e-5 * abs(want_loss)
    np.testing.assert_allclose(tr.last_coords.cpu().numpy(), want_coords.numpy(), rtol=0, atol=2e-4)
    np.testing.assert_allclose(tr.last_maps.cpu().numpy(), want_maps.numpy(), rtol=0, atol=1e-3 * float(want_maps.abs().max()))
    # every backward launch of the step, recomputed in float64 from its own inputs
    assert len(chk.wgrad) == 306 and len(chk.dgrad) == 305 and len(chk.bn) >= 300
    worst = chk.worst()
    print('launch-local worst relative errors:', worst)
    assert worst['wgrad'] < 5e-6 and worst['dgrad'] < 2e-5 and worst['bn'] < 5e-6, worst
    # end to end against the oracle: limited by ReLU ties (~1e2 of 3e7 gates differ)
    gl2, cos, med = gradient_agreement(dict(net.named_parameters()), orc.grads())
    print('end-to-end gradient agreement: rel-L2 %.2e cosine %.6f median per-tensor rel-L2 %.2e' % (gl2, cos, med))
    assert cos > 0.999 and gl2 < 5e-2, (gl2, cos, med)


def test_cross_ratio_term_in_the_step_vs_oracle():
    """Tiny 33-joint coordinate model with the cross-ratio term on (weight 0.05, 'bbox12'
    lines, target 4/3): loss and gradients against the oracle, whose loss restatement is
    pinned to the reference's (tests/golden/cr_loss.npz).  The term only counts once
    apply_cr_loss is set (trainer.py:168-169)."""
    from egonet_amd.common.img_proc import get_cr_indices
    cfg = configs.tiny_config('coordinates', num_joints=33)
    net, sd = _tiny_model(cfg, seed=9)
    gen = torch.Generator().manual_seed(4)
    x = synth.synth_crops(3, 3, 64, 64, seed=8)
    tgt = torch.rand(3, 33, 16, 16, generator=gen)
    jt = torch.rand(3, 33, 2, generator=gen) * 64
    # untrained head: coordinates in 0.15..0.86; threshold 0.05 keeps 31 of the 36 lines
    cr = dict(w_cr=0.05, cr_indices=get_cr_indices(), cr_loss_thres=0.05)
    orc = HRNetTrainOracle(sd, cfg, lr=1e-3, cr=cr)
    want_loss, _, want_coords = orc.step(x, tgt, jt, update=False)
    plain = HRNetTrainOracle(sd, cfg, lr=1e-3).step(x, tgt, jt, update=False)[0]
    assert abs(want_loss - plain) > 1e-3 * abs(plain)                     # the term is live in this case
    tr = HRNetTrainStep(net, lr=1e-3, w_cr=0.05, cr_loss_thres=0.05)
    off = float(tr.step(x.cuda(), tgt.cuda(), jt.cuda(), update=False).item())
    assert abs(off - plain) < 2e-5 * abs(plain)                           # first epoch: term off
    tr.apply_cr_loss = True
    loss = float(tr.step(x.cuda(), tgt.cuda(), jt.cuda(), update=False).item())
    # train-mode BatchNorm over 3 x 4 x 4 values in the coordinate head amplifies fp32 noise
    got_coords = tr.last_coords.cpu()
    assert float((got_coords - want_coords).abs().max()) < 2e-4, float((got_coords - want_coords).abs().max())
    # the term alone, on the coordinates the device produced (pinned loss restatement)
    from oracle.hrnet_train_oracle import cross_ratio_loss
    want_term = 0.05 * float(cross_ratio_loss(got_coords, get_cr_indices(), 4 / 3, 0.05, 'sl1'))
    assert abs((loss - off) - want_term) < 1e-3 * want_term + 1e-6, (loss - off, want_term)
    assert abs(loss - want_loss) < 2e-3 * abs(want_loss), (loss, want_loss)
    gl2, cos, med = gradient_agreement(dict(net.named_parameters()), orc.grads())
    assert cos > 0.9999 and gl2 < 1e-2 and med < 5e-3, (gl2, cos, med)


def test_pedestrian_shape_non_square_step_vs_oracle():
    """192 x 256 (W x H) crops like KITTI_train_IGRs_Ped.yml: 48 x 64 maps and a 4 x 3 'valid'
    final conv in the coordinate head (hrnet.py:457-460) -- forward, 12-tap weight gradient and
    the data gradient as a GEMM over the whole 4 x 3 map."""
    cfg = configs.tiny_config('coordinates', input_size=(192, 256))
    net, sd = _tiny_model(cfg, seed=13)
    assert tuple(net.head2[4].weight.shape[2:]) == (4, 3)
    gen = torch.Generator().manual_seed(6)
    x = synth.synth_crops(2, 3, 256, 192, seed=12)
    tgt = torch.rand(2, 5, 64, 48, generator=gen)
    jt = torch.rand(2, 5, 2, generator=gen) * torch.tensor([192.0, 256.0])
    orc = HRNetTrainOracle(sd, cfg, lr=1e-3)
    want_loss, want_maps, want_coords = orc.step(x, tgt, jt, update=False)
    tr = HRNetTrainStep(net, lr=1e-3)
    with LayerChecks(tr) as chk:
        loss = float(tr.step(x.cuda(), tgt.cuda(), jt.cuda(), update=False).item())
    assert abs(loss - want_loss) < 5e-5 * abs(want_loss), (loss, want_loss)
    np.testing.assert_allclose(tr.last_maps.cpu().numpy(), want_maps.numpy(), rtol=0, atol=2e-4)
    assert float((tr.last_coords.cpu() - want_coords).abs().max()) < 2e-4
    worst = chk.worst()
    assert worst['wgrad'] < 2e-5 and worst['dgrad'] < 2e-5 and worst['bn'] < 1e-4, worst
    assert any(s[5] == 4 and e < 2e-5 for e, s in chk.dgrad)        # the 4x3 layer went through the check
    gl2, cos, med = gradient_agreement(dict(net.named_parameters()), orc.grads())
    assert cos > 0.999 and med < 5e-3, (gl2, cos, med)
