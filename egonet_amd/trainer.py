"""Training loops with the reference's signatures (``libs/trainer/trainer.py``:
``train`` :127-263, ``train_cascade`` :25-71, ``get_loader`` :112-125) around the
native training steps.

What is the reference's and stays on the host: the ``DataLoader``, epochs, the
learning-rate schedule (``MultiStepLR``, optimizer.py:8-40), logging, optional
metric callbacks, snapshots.  What is replaced: the five lines of its hot loop
(trainer.py:183-209, ``zero_grad / forward / loss / backward / step``) -- one call
of ``HRNetTrainStep.step`` / ``LifterTrainStep.step`` (HIP launches only).

``evaluate`` (trainer.py:395-513) is the reference's validation loop: ``model.eval()`` and
``model(data)`` with autograd enabled -- which here runs the HIP inference program (the modules
ignore the autograd mode in eval mode) -- then the caller's loss / evaluator objects.

``loss_func`` and ``optim`` are accepted like in the reference; they are read, not
executed: the loss weights come from ``loss_func.comp_dict`` when it has one
(JointsCompositeLoss, function.py:61-93) else from
``cfgs['heatmapModel']['loss_weight_list']``; the learning rate of every epoch is
read from ``optim.param_groups`` after ``sche.step()``.  Plotting and debug-image
dumps of the reference are not reproduced.
"""
import gc
import os
import time

import numpy as np
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
    params = [p for p in model.parameters() if p.requires_grad]
    wd = float(o.get('weight_decay', 0.0) or 0.0)
    if o['optim_type'] == 'adam':
        optim = torch.optim.Adam(params, lr=o['lr'], weight_decay=wd)
    elif o['optim_type'] == 'sgd':
        optim = torch.optim.SGD(params, lr=o['lr'], momentum=float(o.get('momentum', 0.0) or 0.0), weight_decay=wd)
    else:
        raise NotImplementedError(o['optim_type'])              # as the reference (optimizer.py:27-28)
    sche = torch.optim.lr_scheduler.MultiStepLR(optim, milestones=o['milestones'], gamma=o['gamma'])
    return optim, sche


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


_HOST_GROUP = None


def _make_host_group():
    """The host-side (gloo) group the waiting ranks use during rank 0's validation, created ONCE at the start of
    ``train`` -- where every rank arrives together (``dist.new_group`` is itself a collective: made lazily at the first
    mid-epoch evaluation, ranks != 0 would sit in it for the whole validation and could time out while rank 0 is still
    evaluating).  False = could not be made (the default group's barrier is used)."""
    global _HOST_GROUP
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    if _HOST_GROUP is None:
        import datetime
        try:
            _HOST_GROUP = dist.new_group(backend='gloo', timeout=datetime.timedelta(hours=12))
        except Exception:
            _HOST_GROUP = False


def _wait_for_rank0():
    """Ranks != 0 wait here while rank 0 validates: a barrier of the host-side (gloo) group (12 h timeout) where it
    could be made -- a GPU barrier kernel would sit on the device for the whole validation and is subject to the NCCL
    watchdog timeout of the training group -- else the default group's barrier."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    _make_host_group()          # (a no-op after train()'s call; callers outside train() make it at their first wait)
    if _HOST_GROUP:
        dist.barrier(group=_HOST_GROUP)
    else:
        dist.barrier()


def _optim_kwargs(optim, cfgs):
    """optimizer family / momentum / weight decay of the native update, from the torch optimizer object
    ``prepare_optim`` made (or, without one, from cfgs['optimizer'])."""
    if optim is not None:
        g = optim.param_groups[0]
        # the native update is ONE launch over one flat buffer with one set of hyper-parameters: parameter
        # groups that differ (per-group weight decay / momentum / betas) cannot be collapsed silently
        keys = ('momentum', 'weight_decay', 'betas', 'eps', 'dampening', 'nesterov', 'amsgrad')
        for other in optim.param_groups[1:]:
            diff = [k for k in keys if other.get(k) != g.get(k)]
            if diff:
                raise NotImplementedError('native update: param_groups with different %s (one flat buffer, one '
                                          'set of hyper-parameters)' % ', '.join(diff))
        if isinstance(optim, torch.optim.SGD):
            if g.get('dampening', 0) or g.get('nesterov', False):
                raise NotImplementedError('SGD with dampening / Nesterov momentum')
            return dict(optim_type='sgd', momentum=float(g.get('momentum', 0.0)), weight_decay=float(g['weight_decay']))
        if isinstance(optim, torch.optim.Adam) and not isinstance(optim, torch.optim.AdamW):
            if g.get('amsgrad', False):
                raise NotImplementedError('Adam with amsgrad')
            return dict(optim_type='adam', betas=tuple(g['betas']), eps=float(g['eps']),
                        weight_decay=float(g['weight_decay']))
        raise NotImplementedError('native update for %s' % type(optim).__name__)
    o = cfgs['optimizer']
    return dict(optim_type=o.get('optim_type', 'adam'), momentum=float(o.get('momentum', 0.0) or 0.0),
                weight_decay=float(o.get('weight_decay', 0.0) or 0.0))


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
    for k in spec[:2]:
        if k not in ('mse', 'l1', 'sl1', 'None'):
            raise NotImplementedError('loss criterion %r (loss_dict knows mse, l1, sl1)' % (k,))
    w_hm = float(wl[0]) if spec[0] != 'None' else 0.0
    w_coor = float(wl[1]) if spec[1] != 'None' else 0.0
    cr = {}
    if spec[0] not in ('mse', 'None'):
        cr['hm_type'] = spec[0]          # keywords of HRNetTrainStep beyond the shipped mse / l1 pair
    if spec[1] not in ('l1', 'None'):
        cr['coor_type'] = spec[1]
    if spec[2] != 'None' and wl[2] not in _OFF:
        if spec[2] not in ('mse', 'l1', 'sl1'):
            raise NotImplementedError('cross-ratio criterion %r' % spec[2])
        cr.update(w_cr=float(wl[2]), cr_type=spec[2],
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
            w_coor, cr = 0.0, {k: v for k, v in cr.items() if k == 'hm_type'}
            # JointsMSELoss(use_target_weight=...) (tools/train_IGRs.py:42, function.py:22-46)
            cr['use_target_weight'] = bool(getattr(loss_func, 'use_target_weight',
                                                   cfgs.get('training_settings', {}).get('use_target_weight', False)))
        cr.update(_optim_kwargs(optim, cfgs))
        step = HRNetTrainStep(inner, lr=lr, w_hm=w_hm, w_coor=w_coor, grad_sync=sync, sigma=sigma, **cr)
        step.apply_cr_loss = bool(getattr(loss_func, 'apply_cr_loss', False))
        return step
    if isinstance(inner, FCmodel.FCModel):
        return LifterTrainStep(inner, lr=lr, grad_sync=sync, **_optim_kwargs(optim, cfgs))
    raise TypeError('no native training step for %s' % type(inner).__name__)


class _Acc(object):
    """The reference's AverageMeter for the optional training metric (utils.py:149-182)."""

    def __init__(self):
        self.sum, self.count, self.others = 0.0, 0, None

    def update(self, val, n=1, others=None):
        self.sum += float(val) * n
        self.count += n
        self.others = others

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0.0


def evaluate(eval_dataset, model, loss_func, cfgs, logger, evaluator, save=False, save_path=None,
             collate_fn=None, epoch=None, sample_num=20):
    """trainer.py:395-513.  ``model.eval()`` + ``model(data)`` per batch with autograd ENABLED, like
    the reference: for CUDA batches that is the HIP inference program (hrnet.py / FCmodel.py
    ``forward``).  ``loss_func`` (the caller's criterion, optional) and ``evaluator``
    (``update(prediction, ground_truth=, meta_data=)`` / ``report(logger)``) are called like the
    reference calls them; 3D plotting (``vis_epoch``) is outside the hot path and skipped.
    Returns the mean validation loss (None without ``loss_func``)."""
    ts = cfgs['testing_settings']
    unnorm = bool(ts.get('unnormalize', False))
    stats = eval_dataset.statistics if unnorm else None
    model.eval()
    if ts.get('apply_dropout', False):
        # trainer.py:424-428: dropout layers back in train mode ("a loss similar to the training loss"); the model's
        # forward sees that state and takes its torch graph (FCModel._dropout_active) -- the HIP program has no dropout
        def apply_dropout(m):
            if type(m) == torch.nn.Dropout:
                m.train()
        model.apply(apply_dropout)
    loader = get_loader(eval_dataset, cfgs, 'testing', collate_fn)
    use_cuda = cfgs.get('use_gpu', True) and torch.cuda.is_available()
    loss_sum, seen = 0.0, 0
    preds, gts = [], []
    for batch_idx, (data, target, weights, meta) in enumerate(loader):
        if use_cuda:
            data, target = data.cuda(), target.cuda()
            weights = weights.cuda() if torch.is_tensor(weights) else weights
        prediction = model(data)
        if loss_func is not None:
            with torch.no_grad():
                loss = loss_func(prediction, target, weights, meta)
            loss_sum += float(loss.item()) * data.size(0)
            seen += data.size(0)
        if unnorm:
            target = eval_dataset.unnormalize(target.data.cpu().numpy(), stats['mean_out'], stats['std_out'])
            prediction = eval_dataset.unnormalize(prediction.data.cpu().numpy(), stats['mean_out'], stats['std_out'])
        if evaluator is not None:
            evaluator.update(prediction, ground_truth=target, meta_data=meta)
        if save:
            preds.append(prediction if isinstance(prediction, np.ndarray) else
                         (prediction[0] if isinstance(prediction, tuple) else prediction).data.cpu().numpy())
            gts.append(target if isinstance(target, np.ndarray) else target.data.cpu().numpy())
    if save:
        np.save(save_path, np.array({'pred': np.concatenate(preds, axis=0), 'error': [],
                                     'gt': np.concatenate(gts, axis=0)}, dtype=object))
    if evaluator is not None:
        evaluator.report(logger)
    mean_loss = loss_sum / seen if seen else None
    if mean_loss is not None:
        logger.info('Validation%s: loss %.6f over %d samples' % (
            '' if epoch is None else ' (epoch %d)' % epoch, mean_loss, seen))
    return mean_loss


def train(train_dataset, model, loss_func, optim, sche, cfgs, logger, metric_func=None, stats=None,
          valid_dataset=None, collate_fn=None, save_debug=False, evaluate_fn=None, evaluator=None):
    """trainer.py:127-263.  Validation during training (``eval_during``): ``evaluate_fn(valid_dataset,
    model, epoch)`` if given, else this module's ``evaluate`` with ``evaluator`` (the reference builds
    its ``Evaluator`` from libs/metric, which is outside this package: pass one in)."""
    ts = cfgs['training_settings']
    total_epochs, report_every = ts['total_epochs'], ts['report_every']
    eval_during = bool(ts.get('eval_during', False)) and valid_dataset is not None
    if eval_during and evaluate_fn is None and evaluator is None:
        logger.warning('training_settings.eval_during is set but neither evaluate_fn nor evaluator was given: '
                       'no validation during training')
        eval_during = False
    if eval_during and evaluate_fn is None:
        def evaluate_fn(ds, mdl, ep):
            return evaluate(ds, mdl, loss_func if callable(loss_func) else None, cfgs, logger, evaluator,
                            collate_fn=collate_fn, epoch=ep)
    eval_every = ts.get('eval_every', 0)
    eval_start = ts.get('eval_start_epoch', 0)
    step = make_step(model, cfgs, loss_func, optim)
    is_hc = isinstance(step, HRNetTrainStep)
    dev = step.dev
    x_buffer, y_buffer = [], []
    frozen = False
    if eval_during:
        _make_host_group()      # all ranks are here together: the group exists before anyone has to wait in it
    try:
        for epoch in range(1, total_epochs + 1):
            if epoch > 1 and is_hc:
                step.apply_cr_loss = True                       # trainer.py:168-169: L_cr from the second epoch on
            model.train()
            if sche is not None:
                sche.step()                                     # trainer.py:177, before the epoch like the reference
            if optim is not None:
                step.lr = optim.param_groups[0]['lr']
            loader = get_loader(train_dataset, cfgs, 'training', collate_fn)
            total_batches, t_epoch = len(loader), time.time()
            acc = _Acc()
            for batch_idx, (data, target, weights, meta) in enumerate(loader):
                data = data.to(dev, non_blocking=True)
                target = target.to(dev, non_blocking=True)
                if is_hc:
                    joints = meta['transformed_joints'] if step.w_coor else None
                    loss = step.step(data, target, joints,
                                     target_weight=weights if step.use_target_weight else None)
                    prediction = (step.last_maps, step.last_coords) if step.last_coords is not None else step.last_maps
                else:
                    loss = step.step(data, target)
                    prediction = None
                # the optional metric runs on EVERY batch and accumulates, like the reference
                # (trainer.py:200-205); its decode is on the device, its result is a host number
                if metric_func is not None and prediction is not None:
                    avg_acc, cnt, others = metric_func(prediction, meta, cfgs)
                    acc.update(avg_acc, n=cnt, others=others)
                if not frozen:
                    # everything built so far (model, datasets, the step's packed filters, torch's module
                    # tables: millions of objects) moves to the permanent generation: the full collections
                    # the per-iteration Python objects trigger every ~10 iterations then take < 1 ms
                    # instead of 35 ms of a host that has 1 500 launches per iteration to issue
                    gc.collect()
                    gc.freeze()
                    frozen = True
                if batch_idx % report_every == 0:               # loss read-back only here
                    lv = float(loss.item())
                    logger.info('Epoch: [%d][%d/%d]  loss %.6f  lr %.2e  %.1f samples/s' % (
                        epoch, batch_idx, total_batches, lv, step.lr,
                        (batch_idx + 1) * data.size(0) / max(time.time() - t_epoch, 1e-9)))
                    if acc.count:
                        logger.info('          metric %.6f (running mean over %d)' % (acc.avg, acc.count))
                    x_buffer.append(total_batches * (epoch - 1) + batch_idx)
                    y_buffer.append(lv)
                if eval_during and epoch > eval_start and batch_idx and eval_every and batch_idx % eval_every == 0:
                    parallel.broadcast_buffers(model, src=0)    # running statistics follow rank 0 (DataParallel semantics)
                    # every rank holds the same weights and (now) buffers: rank 0 evaluates the validation set
                    # (ts['eval_all_ranks'] restores one evaluation per rank), the others wait for it.
                    # CONTRACT: unless 'eval_all_ranks' is set, `evaluate_fn` runs on rank 0 ONLY -- it must
                    # not contain collectives or DistributedSampler logic (they would wait for ranks that never call
                    # them).  The waiting ranks block in a barrier of a HOST-side gloo group made at the start of
                    # train() (12 h timeout), not on a GPU barrier kernel; if that group could not be made, in the
                    # default group's barrier (bounded by ITS timeout: choose `timeout=` in init_process_group).
                    if _rank() == 0 or ts.get('eval_all_ranks', False):
                        evaluate_fn(valid_dataset, model, epoch)
                    _wait_for_rank0()
                    model.train()
            if epoch in ts.get('snapshot_epochs', []):
                parallel.broadcast_buffers(model, src=0)
                out_dir = cfgs.get('dirs', {}).get('output', '.')
                path = os.path.join(out_dir, '%s_%d.pth' % (cfgs.get('exp_type', 'model'), epoch))
                if _rank() == 0:       # one writer: concurrent truncating writers can leave a torn checkpoint
                    logger.info('=> Snapshot model to {}'.format(path))
                    inner = model.module if hasattr(model, 'module') else model
                    tmp = path + '.tmp'
                    torch.save(inner.state_dict(), tmp)
                    os.replace(tmp, path)
                _barrier()             # nobody reads / resumes from the file before it is complete
    finally:
        if frozen:
            gc.unfreeze()
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
