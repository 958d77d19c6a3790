"""Training loops with the reference's signatures (``libs/trainer/trainer.py``:
``train`` :127-263, ``train_cascade`` :25-71, ``get_loader`` :112-125) around the
native training steps.

What is the reference's and stays on the host: the ``DataLoader``, epochs, the
learning-rate schedule (``MultiStepLR``, optimizer.py:8-40), logging, optional
metric callbacks, snapshots.  What is replaced: the five lines of its hot loop
(trainer.py:183-209, ``zero_grad / forward / loss / backward / step``) -- one call
of ``HRNetTrainStep.step`` / ``LifterTrainStep.step`` (HIP launches only).

``loss_func`` and ``optim`` are accepted like in the reference; they are read, not
executed: the loss weights come from ``loss_func.comp_dict`` when it has one
(JointsCompositeLoss, function.py:61-93) else from
``cfgs['heatmapModel']['loss_weight_list']``; the learning rate of every epoch is
read from ``optim.param_groups`` after ``sche.step()``.  Plotting and debug-image
dumps of the reference are not reproduced.
"""
import os
import time

import torch

from . import parallel
from .model import FCmodel
from .model.heatmapModel.hrnet import PoseHighResolutionNet
from .train_hrnet import HRNetTrainStep
from .train_lifter import LifterTrainStep


def get_loader(dataset, cfgs, split, collate_fn=None):
    setting = cfgs[split + '_settings']
    kw = dict(batch_size=setting['batch_size'], num_workers=setting['num_threads'], shuffle=setting['shuffle'])
    if collate_fn is not None:
        kw['collate_fn'] = collate_fn
    return torch.utils.data.DataLoader(dataset, **kw)


def prepare_optim(model, cfgs):
    """optimizer.py:8-40 -- the torch objects carry the schedule; the native step does the update."""
    o = cfgs['optimizer']
    if o['optim_type'] != 'adam':
        raise NotImplementedError('native training implements Adam (the shipped configs), got %r' % o['optim_type'])
    if o.get('weight_decay', 0.0):
        raise NotImplementedError('weight_decay != 0')
    params = [p for p in model.parameters() if p.requires_grad]
    optim = torch.optim.Adam(params, lr=o['lr'], weight_decay=0.0)
    sche = torch.optim.lr_scheduler.MultiStepLR(optim, milestones=o['milestones'], gamma=o['gamma'])
    return optim, sche


_CRITERION_NAMES = {'MSELoss': 'mse', 'L1Loss': 'l1', 'SmoothL1Loss': 'sl1'}      # loss_dict, function.py:17-20
_OFF = (None, 'None', 0, 0.0)


def _loss_weights(loss_func, cfgs):
    """(w_hm, w_coor, cross-ratio keywords of HRNetTrainStep) from a JointsCompositeLoss-like
    object (comp_dict / cr_indices / target_cr / cr_loss_thres, function.py:61-93 and
    train_IGRs.py:33-46) or, without one, from cfgs['heatmapModel']."""
    hm = cfgs.get('heatmapModel', {})
    comp = getattr(loss_func, 'comp_dict', None)
    if comp is not None:
        spec = [_CRITERION_NAMES.get(type(comp[k][0]).__name__, type(comp[k][0]).__name__) if k in comp else 'None'
                for k in ('hm', 'coor', 'cr')]
        wl = [comp[k][1] if k in comp else 'None' for k in ('hm', 'coor', 'cr')]
    else:
        wl = list(hm.get('loss_weight_list', [1.0, 0.1, 'None']))
        spec = list(hm.get('loss_spec_list', ['mse', 'l1', 'None']))
    if spec[0] not in ('mse', 'None') or spec[1] not in ('l1', 'None'):
        raise NotImplementedError('native loss: heat-map term mse, coordinate term l1 (the shipped configs); got %r'
                                  % (spec[:2],))
    w_hm = float(wl[0]) if spec[0] != 'None' else 0.0
    w_coor = float(wl[1]) if spec[1] != 'None' else 0.0
    cr = {}
    if spec[2] != 'None' and wl[2] not in _OFF:
        if spec[2] not in ('mse', 'l1', 'sl1'):
            raise NotImplementedError('cross-ratio criterion %r' % spec[2])
        cr = dict(w_cr=float(wl[2]), cr_type=spec[2],
                  cr_indices=getattr(loss_func, 'cr_indices', None),
                  target_cr=getattr(loss_func, 'target_cr', None) or 4.0 / 3.0,
                  cr_loss_thres=getattr(loss_func, 'cr_loss_thres', hm.get('cr_loss_threshold', 0.15)))
    return w_hm, w_coor, cr


def make_step(model, cfgs, loss_func=None, optim=None):
    """The native step object for ``model`` (HC or L), configured like the reference's loss / optimizer."""
    lr = optim.param_groups[0]['lr'] if optim is not None else cfgs['optimizer']['lr']
    sync = parallel.FlatGradSync() if torch.distributed.is_available() and torch.distributed.is_initialized() \
        and torch.distributed.get_world_size() > 1 else None
    inner = model.module if hasattr(model, 'module') else model          # DataParallel / DDP wrappers
    if isinstance(inner, PoseHighResolutionNet):
        w_hm, w_coor, cr = _loss_weights(loss_func, cfgs)
        sigma = cfgs.get('heatmapModel', {}).get('sigma', 1)
        if inner.head_type == 'heatmap':
            w_coor, cr = 0.0, {}
        step = HRNetTrainStep(inner, lr=lr, w_hm=w_hm, w_coor=w_coor, grad_sync=sync, sigma=sigma, **cr)
        step.apply_cr_loss = bool(getattr(loss_func, 'apply_cr_loss', False))
        return step
    if isinstance(inner, FCmodel.FCModel):
        return LifterTrainStep(inner, lr=lr, grad_sync=sync)
    raise TypeError('no native training step for %s' % type(inner).__name__)


def train(train_dataset, model, loss_func, optim, sche, cfgs, logger, metric_func=None, stats=None,
          valid_dataset=None, collate_fn=None, save_debug=False, evaluate_fn=None):
    """trainer.py:127-263.  ``evaluate_fn(valid_dataset, model, epoch)`` (optional) stands in for the
    reference's ``evaluate`` call during training (``eval_during``)."""
    ts = cfgs['training_settings']
    total_epochs, report_every = ts['total_epochs'], ts['report_every']
    eval_during = ts.get('eval_during', False) and valid_dataset is not None and evaluate_fn is not None
    eval_every = ts.get('eval_every', 0)
    eval_start = ts.get('eval_start_epoch', 0)
    step = make_step(model, cfgs, loss_func, optim)
    is_hc = isinstance(step, HRNetTrainStep)
    dev = step.dev
    x_buffer, y_buffer = [], []
    for epoch in range(1, total_epochs + 1):
        if epoch > 1 and is_hc:
            step.apply_cr_loss = True                       # trainer.py:168-169: L_cr from the second epoch on
        model.train()
        if sche is not None:
            sche.step()                                     # trainer.py:177, before the epoch like the reference
        if optim is not None:
            step.lr = optim.param_groups[0]['lr']
        loader = get_loader(train_dataset, cfgs, 'training', collate_fn)
        total_batches, seen, loss_sum, t_epoch = len(loader), 0, 0.0, time.time()
        for batch_idx, (data, target, weights, meta) in enumerate(loader):
            data = data.to(dev, non_blocking=True)
            target = target.to(dev, non_blocking=True)
            if is_hc:
                joints = meta['transformed_joints'] if step.w_coor else None
                loss = step.step(data, target, joints)
                prediction = (step.last_maps, step.last_coords) if step.last_coords is not None else step.last_maps
            else:
                loss = step.step(data, target)
                prediction = None
            if batch_idx % report_every == 0:               # the only host sync of the loop
                lv = float(loss.item())
                seen += data.size(0)
                loss_sum += lv
                logger.info('Epoch: [%d][%d/%d]  loss %.6f  lr %.2e  %.1f samples/s' % (
                    epoch, batch_idx, total_batches, lv, step.lr,
                    (batch_idx + 1) * data.size(0) / max(time.time() - t_epoch, 1e-9)))
                x_buffer.append(total_batches * (epoch - 1) + batch_idx)
                y_buffer.append(lv)
                if metric_func is not None and prediction is not None:
                    metric_func(prediction, meta, cfgs)
            if eval_during and epoch > eval_start and batch_idx and eval_every and batch_idx % eval_every == 0:
                evaluate_fn(valid_dataset, model, epoch)
                model.train()
        if epoch in ts.get('snapshot_epochs', []):
            out_dir = cfgs.get('dirs', {}).get('output', '.')
            path = os.path.join(out_dir, '%s_%d.pth' % (cfgs.get('exp_type', 'model'), epoch))
            logger.info('=> Snapshot model to {}'.format(path))
            inner = model.module if hasattr(model, 'module') else model
            torch.save(inner.state_dict(), path)
    logger.info('Training finished.')
    return {'model': model, 'batch_idx': x_buffer, 'loss': y_buffer}


def train_cascade(train_dataset, valid_dataset, cfgs, logger):
    """trainer.py:25-71: the lifter sub-model(s) L.pth, one stage after the other."""
    cascade = FCmodel.get_cascade()
    stage_record = []
    for stage_id in range(cfgs['cascade']['num_stages']):
        input_size, output_size = train_dataset.get_input_output_size()
        cfgs['FCModel']['input_size'] = input_size
        cfgs['FCModel']['output_size'] = output_size
        stage_model = FCmodel.get_fc_model(stage_id + 1, cfgs=cfgs, input_size=input_size, output_size=output_size)
        if not cfgs.get('use_gpu', True):
            raise ValueError('native training runs on the GPU (use_gpu: true)')
        stage_model = stage_model.cuda()
        optim, sche = prepare_optim(stage_model, cfgs)
        record = train(train_dataset=train_dataset, valid_dataset=valid_dataset, model=stage_model, loss_func=None,
                       optim=optim, sche=sche, stats=None, cfgs=cfgs, logger=logger)
        stage_record.append((record['batch_idx'], record['loss']))
        cascade.append(record['model'].cpu())
    return {'cascade': cascade, 'record': stage_record}
