"""The reference-shaped training loops (egonet_amd/trainer.py: ``train``,
``train_cascade`` of libs/trainer/trainer.py) on synthetic datasets: DataLoader,
schedule and logging on the host, every iteration a native step."""
import logging

import numpy as np
import pytest
import torch

from egonet_amd import configs, synth, trainer
from egonet_amd.model import FCmodel
from egonet_amd.model.heatmapModel import hrnet

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_autotune(monkeypatch):
    monkeypatch.setenv('EGONET_AMD_AUTOTUNE', '0')


class _Lines(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def _logger():
    lg = logging.getLogger('egonet_amd.test_trainer')
    lg.setLevel(logging.INFO)
    h = _Lines()
    lg.handlers = [h]
    return lg, h


class _LiftSet(torch.utils.data.Dataset):
    """2D -> 3D pairs from a fixed linear map (what train_lifting's dataset yields:
    data, target, weights, meta)."""

    def __init__(self, n=512):
        g = torch.Generator().manual_seed(0)
        w = torch.randn(10, 12, generator=g) * 0.5          # drawn first: the same map for every n
        self.x = torch.randn(n, 10, generator=g)
        self.y = self.x @ w

    def get_input_output_size(self):
        return 10, 12

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], self.y[i], torch.ones(1), {'idx': i}


def _train_cfg(cfg, epochs, batch, report=2):
    cfg = configs.clone(cfg)
    cfg.update(use_gpu=True, exp_type='test', cascade={'num_stages': 1},
               optimizer={'optim_type': 'adam', 'lr': 5e-3, 'weight_decay': 0.0, 'momentum': 0.9,
                          'milestones': [3], 'gamma': 0.5},
               training_settings={'total_epochs': epochs, 'batch_size': batch, 'num_threads': 0, 'shuffle': False,
                                  'report_every': report, 'eval_during': False, 'plot_loss': False})
    return cfg


def test_train_cascade_trains_the_lifter_and_returns_a_cpu_cascade():
    cfg = _train_cfg(configs.tiny_config(), epochs=6, batch=64)
    cfg['FCModel']['dropout'] = 0.0
    lg, h = _logger()
    out = trainer.train_cascade(_LiftSet(), None, cfg, lg)
    (idx, losses), = out['record']
    assert len(losses) == 6 * 4 and idx[:3] == [0, 2, 4]            # 8 batches per epoch, report every 2
    assert losses[-1] < 0.5 * losses[0]
    model = out['cascade'][0]
    assert not next(model.parameters()).is_cuda                      # train_lifting.py:51 saves cascade[0].cpu()
    sd = model.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
    fresh = FCmodel.get_fc_model(1, cfg, 10, 12)
    fresh.load_state_dict(sd)                                        # L.pth round trip
    assert int(sd['batch_norm1.num_batches_tracked']) == 48
    assert 'lr 5.00e-03' in h.lines[0] and 'lr 2.50e-03' in h.lines[-2]          # MultiStepLR milestone 3 (early, as in the reference)
    assert h.lines[-1] == 'Training finished.'


class _CropSet(torch.utils.data.Dataset):
    def __init__(self, n=8):
        g = torch.Generator().manual_seed(1)
        self.x = synth.synth_crops(n, 3, 64, 64, seed=3)
        self.t = torch.rand(n, 5, 16, 16, generator=g)
        self.j = torch.rand(n, 5, 3, generator=g) * 64

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], self.t[i], torch.ones(5, 1), {'transformed_joints': self.j[i].numpy()}


def test_train_runs_the_hc_loop_with_metric_callback_and_snapshot(tmp_path):
    cfg = _train_cfg(configs.tiny_config('coordinates'), epochs=2, batch=4, report=1)
    cfg['training_settings']['snapshot_epochs'] = [2]
    cfg['dirs'] = {'output': str(tmp_path)}
    net = hrnet.get_pose_net(cfg, is_train=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=9))
    net = net.cuda()
    before = net.conv1.weight.detach().clone()
    optim, sche = trainer.prepare_optim(net, cfg)
    seen = []

    def metric(prediction, meta, cfgs):              # (avg_acc, cnt, others) like get_distance_src
        maps, coords = prediction
        seen.append((tuple(maps.shape), tuple(coords.shape), tuple(meta['transformed_joints'].shape)))
        return 0.25 * len(seen), len(maps), None

    lg, h = _logger()
    rec = trainer.train(_CropSet(), net, None, optim, sche, cfg, lg, metric_func=metric)
    assert len(rec['loss']) == 4 and all(np.isfinite(rec['loss']))
    assert seen[0] == ((4, 5, 16, 16), (4, 5, 2), (4, 5, 3)) and len(seen) == 4     # every batch
    assert any('metric 0.375000 (running mean over 8)' in l for l in h.lines)         # epoch 1: (0.25*4 + 0.5*4) / 8
    assert not torch.equal(net.conv1.weight.detach(), before)
    snap = torch.load(str(tmp_path / 'test_2.pth'))
    assert list(snap) == list(net.state_dict()) and int(snap['bn1.num_batches_tracked']) == 4
    fresh = hrnet.get_pose_net(cfg, is_train=False)
    fresh.load_state_dict(snap)                                      # HC.pth layout
    # the cross-ratio term: its 'bbox12' lines need the 33-joint model; other criteria are refused
    cfg2 = configs.clone(cfg)
    cfg2['heatmapModel']['loss_spec_list'] = ['mse', 'l1', 'sl1']
    cfg2['heatmapModel']['loss_weight_list'] = [1.0, 0.1, 0.01]
    with pytest.raises(ValueError):
        trainer.make_step(net, cfg2)                                 # 5 joints: indices out of range
    cfg3 = _train_cfg(configs.tiny_config('coordinates', num_joints=33), epochs=1, batch=4, report=1)
    cfg3['heatmapModel'].update(loss_spec_list=['mse', 'l1', 'sl1'], loss_weight_list=[1.0, 0.1, 0.01],
                                cr_loss_threshold=0.1)
    net33 = hrnet.get_pose_net(cfg3, is_train=False).cuda()
    step = trainer.make_step(net33, cfg3)
    assert step.w_cr == 0.01 and step.cr_loss_thres == 0.1 and step.cr_idx.shape == (12, 4)
    assert step.apply_cr_loss is False                               # the trainer switches it on in epoch 2
    cfg2['heatmapModel']['loss_spec_list'] = ['sl1', 'mse', 'None']       # any pair of loss_dict (function.py:17-20)
    cfg2['heatmapModel']['loss_weight_list'] = [1.0, 0.1, 'None']
    step2 = trainer.make_step(net, cfg2)
    assert (step2.hm_crit, step2.coor_crit) == (2, 0)
    cfg2['heatmapModel']['loss_spec_list'] = ['huber', 'l1', 'None']
    with pytest.raises(NotImplementedError):
        trainer.make_step(net, cfg2)


def test_train_cascade_with_a_ragged_tail_batch():
    """len(dataset) % batch_size is anything with the reference's DataLoader (no drop_last,
    trainer.py:113-125): 333 = 5 x 64 + 13."""
    cfg = _train_cfg(configs.tiny_config(), epochs=2, batch=64)
    cfg['FCModel']['dropout'] = 0.0
    lg, h = _logger()
    out = trainer.train_cascade(_LiftSet(333), None, cfg, lg)
    (idx, losses), = out['record']
    assert len(losses) == 2 * 3 and all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert int(out['cascade'][0].state_dict()['batch_norm1.num_batches_tracked']) == 12


class _Evaluator(object):
    def __init__(self):
        self.n, self.err = 0, 0.0

    def update(self, prediction, ground_truth=None, meta_data=None):
        p = prediction if isinstance(prediction, np.ndarray) else prediction.detach().cpu().numpy()
        g = ground_truth if isinstance(ground_truth, np.ndarray) else ground_truth.cpu().numpy()
        self.n += len(p)
        self.err += float(np.abs(p - g).sum())

    def report(self, logger):
        logger.info('MAE %.6f over %d' % (self.err / max(self.n, 1) / 12, self.n))


def test_evaluate_runs_the_hip_program_like_the_reference_loop():
    """trainer.evaluate (reference trainer.py:395-513): model.eval() and model(data) with autograd on --
    the HIP program (launch counter), the caller's criterion and evaluator; and eval_during inside
    train() uses it between native steps with the weights of that moment."""
    from egonet_amd import _lib
    L = _lib.lib()
    cfg = _train_cfg(configs.tiny_config(), epochs=1, batch=64)
    cfg['FCModel']['dropout'] = 0.0
    cfg['testing_settings'] = {'batch_size': 100, 'num_threads': 0, 'shuffle': False, 'unnormalize': False,
                               'apply_dropout': False}
    net = FCmodel.get_fc_model(1, cfg, 10, 12)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=3))
    net = net.cuda().train()
    ds = _LiftSet(250)
    lg, h = _logger()
    ev = _Evaluator()
    crit = torch.nn.MSELoss()
    c0 = L.egn_launch_count()
    loss = trainer.evaluate(ds, net, lambda p, t, w, m: crit(p, t), cfg, lg, ev)
    assert L.egn_launch_count() - c0 == 3 * 7 and not net.training          # 3 batches x (relayout + 6 GEMMs)
    assert ev.n == 250 and any(l.startswith('MAE') for l in h.lines)
    want = float(crit(net.cpu().eval()(ds.x), ds.y))                         # torch on the CPU, same weights
    assert abs(loss - want) < 1e-5 * max(1.0, want)
    # eval_during: validation between native steps sees the updated weights
    net = net.cuda()
    cfg['training_settings'].update(eval_during=True, eval_every=2, eval_start_epoch=0, total_epochs=2)
    optim, sche = trainer.prepare_optim(net, cfg)
    lg, h = _logger()
    trainer.train(_LiftSet(512), net, lambda p, t, w, m: crit(p, t), optim, sche, cfg, lg, valid_dataset=ds,
                  evaluator=_Evaluator())
    vals = [float(l.split('loss ')[1].split()[0]) for l in h.lines if l.startswith('Validation')]
    assert len(vals) == 6 and vals[-1] < vals[0]                             # batches 2, 4, 6 of both epochs
    lg, h = _logger()
    trainer.train(_LiftSet(128), net, None, optim, sche, cfg, lg, valid_dataset=ds)      # nothing to evaluate with
    assert any('no validation during training' in l for l in h.lines)
