"""Native training step of the FC lifter (BASELINE config 3) on MI355X:
forward + backward + Adam as HIP launches, against (a) the reference's own
three iterations (tests/golden/lifter_train.npz) and (b) the CPU training
oracle at full size.

Linear biases that feed a BatchNorm have an analytically ZERO gradient (the
batch mean removes them); what is left is rounding noise that Adam normalises
to +-lr steps of random sign -- in the reference too.  Those entries do not
influence the network function and are compared with a tolerance of 3 steps
of lr; everything else is compared tightly."""
import numpy as np
import pytest
import torch

from conftest import golden
from egonet_amd import configs, synth
from egonet_amd.model import FCmodel
from egonet_amd.train_lifter import LifterTrainStep
from oracle.lifter_train_oracle import LifterTrainOracle

pytestmark = pytest.mark.gpu


def _is_dead_bias(k):
    return k.endswith('.bias') and not k.startswith('w2.') and 'batch_norm' not in k


def test_lifter_train_step_vs_reference_iterations():
    g = golden('lifter_train.npz')
    sd0 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd0/')}
    sd3 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd3/')}
    cfg = configs.tiny_config()
    cfg['FCModel']['dropout'] = 0.0
    net = FCmodel.get_fc_model(1, cfg, 10, 12)
    net.load_state_dict(sd0)
    net = net.cuda().train()
    tr = LifterTrainStep(net, lr=1e-3)
    losses = []
    for i in range(3):
        loss = tr.step(torch.from_numpy(g['xs'][i]).cuda(), torch.from_numpy(g['ys'][i]).cuda())
        losses.append(float(loss.item()))
    np.testing.assert_allclose(losses, g['losses'], rtol=2e-5)
    got = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for k, v in sd3.items():
        tol = 3.5e-3 if _is_dead_bias(k) else 2e-5
        if k.endswith('running_mean'):
            tol = 1e-3          # the batch mean of z = a W^T + b carries the randomly walking dead bias b
        if k.endswith('num_batches_tracked'):
            assert int(got[k]) == int(v) == 3
            continue
        d = np.abs(got[k].numpy() - v.numpy())
        # three Adam steps: an entry whose gradient is rounding noise moves by +-lr-scaled amounts in either direction, and
        # which way depends on the summation order of the kernel the tuner picked (the K-split FC kernel, cfg 79, sums
        # in four interleaved quarters): all but a handful of entries within tol, none beyond 3 tol
        assert d.max() <= 3 * tol and np.mean(d > tol) < 1e-3, (k, float(d.max()), float(np.mean(d > tol)))


@pytest.mark.parametrize('leaky,optim', [
    (True, dict(optim_type='adam', weight_decay=0.0)),
    (True, dict(optim_type='sgd', momentum=0.9, weight_decay=1e-3)),
    (False, dict(optim_type='adam', weight_decay=1e-2)),
])
def test_lifter_leaky_relu_and_optimizer_variants_vs_oracle(leaky, optim):
    """FCModel(leaky=True) trains with nn.LeakyReLU() (FCmodel.py:19-22); optimizer.py:8-40 offers SGD with
    momentum and weight decay for both optimizers: three steps against the CPU oracle."""
    cfg = configs.tiny_config()
    cfg['FCModel'].update(dropout=0.0, leaky=leaky)
    net = FCmodel.get_fc_model(1, cfg, 10, 12)
    assert net.leaky == leaky
    sd = synth.synth_state_dict(net.state_dict(), seed=4)
    net.load_state_dict(sd)
    net = net.cuda().train()
    lr = 1e-2 if optim['optim_type'] == 'sgd' else 1e-3
    num_blocks = len(net.res_blocks)
    orc = LifterTrainOracle(sd, lr=lr, num_blocks=num_blocks, leaky=leaky, optim=optim)
    tr = LifterTrainStep(net, lr=lr, **optim)
    g = torch.Generator().manual_seed(5)
    for i in range(3):
        x, y = torch.randn(37, 10, generator=g), torch.randn(37, 12, generator=g)
        want = orc.step(x, y)
        loss = tr.step(x.cuda(), y.cuda())
        assert abs(float(loss.item()) - want) < 5e-5 * abs(want), (i, float(loss.item()), want)
        if i == 0:
            named = dict(net.named_parameters())
            for k, gv in orc.grads().items():
                if _is_dead_bias(k):
                    continue
                np.testing.assert_allclose(named[k].grad.cpu().numpy(), gv.numpy(), rtol=0,
                                           atol=2e-5 * max(1.0, float(gv.abs().max())), err_msg=k)
    got = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for k in orc.param_keys:
        tol = (3.5 * lr if _is_dead_bias(k) and optim['optim_type'] == 'adam' else 5e-5)
        np.testing.assert_allclose(got[k].numpy(), orc.sd[k].detach().numpy(), rtol=0, atol=tol, err_msg=k)


@pytest.mark.parametrize('batch', [256, 4096])          # 4096 = BASELINE config 3's batch
def test_lifter_gradients_full_size_vs_oracle(batch):
    cfg = configs.w48_config()
    cfg['FCModel']['dropout'] = 0.0
    net = FCmodel.get_fc_model(1, cfg, 66, 96)
    sd = synth.synth_state_dict(net.state_dict(), seed=2)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(batch, 66, generator=g), torch.randn(batch, 96, generator=g)
    orc = LifterTrainOracle(sd, lr=1e-3)
    want_loss = orc.step(x, y)
    want = orc.grads()
    net = net.cuda().train()
    tr = LifterTrainStep(net, lr=1e-3)
    loss = tr.step(x.cuda(), y.cuda(), update=False)
    assert abs(float(loss.item()) - want_loss) < 1e-5 * max(1.0, want_loss)
    named = dict(net.named_parameters())
    if batch == 4096:
        # 4 M pre-activations per layer: a handful sit within rounding of zero and their ReLU gates fall
        # differently in ANY two fp32 implementations (torch's fp32 oracle is 1e-3 .. 7e-3 of the largest
        # entry away from float64 too) -- compare with the FLOAT64 oracle in the L2 sense, per tensor
        o64 = LifterTrainOracle({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()},
                                lr=1e-3)
        o64.step(x.double(), y.double())
        for k, g64 in o64.grads().items():
            if _is_dead_bias(k):
                assert float(named[k].grad.abs().max()) < 1e-5
                continue
            got, ref32 = named[k].grad.cpu().double(), want[k].double()
            rel = float((got - g64).norm() / g64.norm())
            rel32 = float((ref32 - g64).norm() / g64.norm())
            assert rel < max(5e-3, 4 * rel32), (k, rel, rel32)
        return
    for k, gw in want.items():
        got = named[k].grad.cpu()
        scale = float(gw.abs().max())
        if _is_dead_bias(k):
            assert float(got.abs().max()) < 1e-5           # analytically zero
            continue
        np.testing.assert_allclose(got.numpy(), gw.numpy(), rtol=0, atol=2e-4 * scale + 1e-8, err_msg=k)
    # running statistics follow torch's momentum / unbiased-variance rule
    np.testing.assert_allclose(net.batch_norm1.running_var.cpu().numpy(),
                               orc.sd['batch_norm1.running_var'].numpy(), rtol=1e-4)
    np.testing.assert_allclose(net.batch_norm1.running_mean.cpu().numpy(),
                               orc.sd['batch_norm1.running_mean'].numpy(), atol=1e-5)


def test_graphed_lifter_step_equals_eager_steps():
    """The whole iteration captured as one hipGraph (egonet_amd/graph.py): replays give
    bit-identical parameters to eager steps -- the Adam step counter lives on the device -- and
    the warm-up iterations of the capture leave no trace (weights, moments, step counter,
    BatchNorm buffers are restored: the first replay is iteration 1)."""
    from egonet_amd.graph import GraphedStep
    cfg = configs.tiny_config()
    cfg['FCModel']['dropout'] = 0.0
    gen = torch.Generator().manual_seed(1)
    xs = torch.randn(6, 32, 10, generator=gen).cuda()
    ys = torch.randn(6, 32, 12, generator=gen).cuda()
    nets = []
    for graphed in (False, True):
        net = FCmodel.get_fc_model(1, cfg, 10, 12)
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=4))
        net = net.cuda().train()
        tr = LifterTrainStep(net, lr=1e-3)
        if graphed:
            before = {k: v.clone() for k, v in net.state_dict().items()}
            side = tr.wgrad_stream
            g = GraphedStep(tr, xs[0], ys[0], warmup=2)          # 2 warm-up iterations on batch 0, undone
            assert tr.flat.t == 0 and float(tr.flat.m.abs().max()) == 0.0
            for k, v in net.state_dict().items():
                assert torch.equal(v, before[k]), k
            losses = [float(g(xs[i], ys[i]).item()) for i in range(1, 6)]
            g.close()
            assert tr.wgrad_stream is side                       # handed back for eager steps
        else:
            losses = [float(tr.step(xs[i], ys[i]).item()) for i in range(1, 6)]
        nets.append((net, losses, tr.flat.t))
    assert nets[0][2] == nets[1][2] == 5
    np.testing.assert_allclose(nets[0][1], nets[1][1], rtol=1e-12)     # the loss sum uses atomics: order varies
    for (k, a), (_, b) in zip(nets[0][0].state_dict().items(), nets[1][0].state_dict().items()):
        assert torch.equal(a, b), k


def test_lifter_training_reduces_loss_with_dropout():
    cfg = configs.tiny_config()
    net = FCmodel.get_fc_model(1, cfg, 10, 12).cuda().train()      # dropout 0.5 active
    tr = LifterTrainStep(net, lr=1e-2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 10, generator=g).cuda()
    w = torch.randn(10, 12, generator=g).cuda()
    y = x @ w
    first = float(tr.step(x, y).item())
    for _ in range(60):
        last = float(tr.step(x, y).item())
    assert np.isfinite(last) and last < 0.5 * first


def test_graphed_step_follows_the_lr_schedule():
    """MultiStepLR changes trainer.lr between iterations (egonet_amd/trainer.py): the replayed graph
    reads the learning rate from device memory, refreshed by GraphedStep.__call__."""
    from egonet_amd.graph import GraphedStep
    cfg = configs.tiny_config()
    cfg['FCModel']['dropout'] = 0.0
    gen = torch.Generator().manual_seed(2)
    xs = torch.randn(4, 32, 10, generator=gen).cuda()
    ys = torch.randn(4, 32, 12, generator=gen).cuda()
    sds = []
    for graphed in (False, True):
        net = FCmodel.get_fc_model(1, cfg, 10, 12)
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=4))
        net = net.cuda().train()
        tr = LifterTrainStep(net, lr=1e-3)
        step = GraphedStep(tr, xs[0], ys[0]) if graphed else tr.step
        for i in range(4):
            if i == 2:
                tr.lr = 2.5e-4          # scheduler milestone
            step(xs[i], ys[i])
        sds.append({k: v.clone() for k, v in net.state_dict().items()})
    for k in sds[0]:
        assert torch.equal(sds[0][k], sds[1][k]), k


@pytest.mark.parametrize('batch', [30, 4097, 7])      # (2-sample BatchNorm is too ill-conditioned to compare)
def test_lifter_step_ragged_batch_sizes(batch, monkeypatch):
    """The reference's DataLoader has no drop_last (trainer.py:113-125): the tail batch of an epoch
    has any size.  Loss and gradients against the CPU oracle.  Autotuning is off: these row counts are in no table, and
    which tile configuration an on-box autotuner happens to pick decides the summation order of the 4097-row column
    sums -- the BatchNorm backward's common-mode error (d beta / N in every dz) times sum(a) over 4 097 post-ReLU rows
    puts the comparison at 2-3e-4 of the largest entry either side of the bound (measured r4: fails with one
    selection, passes with the cost model's, with and without this round's epilogue fusions)."""
    monkeypatch.setenv('EGONET_AMD_AUTOTUNE', '0')
    cfg = configs.tiny_config()
    cfg['FCModel']['dropout'] = 0.0
    net = FCmodel.get_fc_model(1, cfg, 10, 12)
    sd = synth.synth_state_dict(net.state_dict(), seed=5)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(batch)
    x, y = torch.randn(batch, 10, generator=g), torch.randn(batch, 12, generator=g)
    orc = LifterTrainOracle(sd, lr=1e-3)
    want_loss = orc.step(x, y)
    want = orc.grads()
    net = net.cuda().train()
    tr = LifterTrainStep(net, lr=1e-3)
    loss = tr.step(x.cuda(), y.cuda(), update=False)
    assert abs(float(loss.item()) - want_loss) < 2e-5 * max(1.0, want_loss)
    named = dict(net.named_parameters())
    for k, gw in want.items():
        if _is_dead_bias(k):
            continue
        scale = float(gw.abs().max())
        np.testing.assert_allclose(named[k].grad.cpu().numpy(), gw.numpy(), rtol=0, atol=3e-4 * scale + 1e-7,
                                   err_msg=k)
    with pytest.raises(ValueError):
        tr.step(x[:1].cuda(), y[:1].cuda())          # BatchNorm1d cannot train on one sample (torch raises too)


def test_lifter_inference_after_native_training_uses_the_updated_weights():
    """Eval-mode forwards between native steps (eval_during, EgoNet.L after fine-tuning) must see the
    weights and BatchNorm statistics the HIP kernels wrote through raw pointers."""
    cfg = configs.tiny_config()
    cfg['FCModel']['dropout'] = 0.0
    net = FCmodel.get_fc_model(1, cfg, 10, 12)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=6))
    net = net.cuda()
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(32, 10, generator=g).cuda(), torch.randn(32, 12, generator=g).cuda()
    y0 = net.eval()(x).clone()
    tr = LifterTrainStep(net.train(), lr=1e-2)
    for _ in range(3):
        tr.step(x, y)
        y1 = net.eval()(x).clone()                   # eval between steps, grad mode on
        net.train()
    assert float((y1 - y0).abs().max()) > 1e-3
    fresh = FCmodel.get_fc_model(1, cfg, 10, 12)
    fresh.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    want = fresh.eval()(x.cpu())                     # torch on the CPU with the trained weights
    np.testing.assert_allclose(y1.cpu().numpy(), want.detach().numpy(), atol=2e-5)
