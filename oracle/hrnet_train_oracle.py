"""CPU oracle: one training iteration of the HRNet heat-map/coordinate model
(TEST INFRASTRUCTURE ONLY).

Restates, on a flat state_dict with torch autograd (functional ops only):
  libs/trainer/trainer.py:183-209    zero_grad -> model(data) -> loss -> backward -> optim.step
  libs/model/heatmapModel/hrnet.py   train-mode forward (oracle/hrnet_oracle.py, BatchNorm
                                     on batch statistics, momentum 0.1)
  libs/loss/function.py:61-202       JointsCompositeLoss with spec ['mse','l1',None]:
        L = w_hm * (1/K) sum_k 0.5 * MSE_mean(hm_k, tgt_k)            (:95-111)
          + w_coor * L1_mean(coords_pred, joints_xy / img_size)       (:155-168, :186-199)
          + w_cr * cross-ratio term (optional)                        (:113-153, :200-202)
     (KITTI_train_IGRs.yml:88-89: weights 1.0 / 0.1, cross-ratio weight 'None' = off)
  libs/optimizer/optimizer.py:8-40   Adam(lr, weight_decay 0)
"""
import torch
import torch.nn.functional as F

from . import hrnet_oracle


_CRIT = {'mse': F.mse_loss, 'l1': F.l1_loss, 'sl1': F.smooth_l1_loss}     # function.py:17-20


def cross_ratio_mask(coords, cr_indices, threshold):
    """function.py:138-153 get_cr_mask: a line is kept when the smallest NON-ZERO entry of
    the 4x4 distance matrix of its points exceeds the threshold (float32, as scipy's
    distance_matrix computes on float32 input)."""
    pts = coords.detach()[:, torch.as_tensor(cr_indices, dtype=torch.long)]          # [N,L,4,2]
    d = (pts[:, :, :, None, :] - pts[:, :, None, :, :]).abs().pow(2).sum(-1).sqrt()
    d = torch.where(d == 0, torch.full_like(d, float('inf')), d)
    m = d.flatten(2).min(-1).values
    return ((m > threshold) & torch.isfinite(m)).float()


def cross_ratio_loss(coords, cr_indices, target_cr=4.0 / 3.0, threshold=0.15, crit='sl1'):
    """function.py:113-136 calc_cross_ratio_loss with img_proc.py:709-720 appro_cr, the
    sample x line double loop vectorised (same fp32 operation order per line)."""
    mask = cross_ratio_mask(coords, cr_indices, threshold)
    if float(mask.sum()) == 0:
        return coords.sum() * 0
    p = coords[:, torch.as_tensor(cr_indices, dtype=torch.long)]                     # [N,L,4,2]
    a, b, c, d = p[:, :, 0], p[:, :, 1], p[:, :, 2], p[:, :, 3]

    def sq(u):
        return (u * u).sum(-1)
    cr = (sq(c - a) * sq(d - b)) / (sq(c - b) * sq(d - a))
    cr = cr / target_cr ** 2
    line = _CRIT[crit](cr, torch.ones_like(cr), reduction='none')
    return (line * mask).sum() / mask.sum()


def joints_mse_loss(maps, target, target_weight=None):
    """JointsMSELoss (libs/loss/function.py:22-46), the criterion of the heat-map head: per joint
    0.5 * mean((pred - gt)^2), with ``use_target_weight`` both maps multiplied by target_weight[:, k] first
    (invisible joints, weight 0, drop out); mean over the joints.  Pinned on the reference's own class
    (tests/golden/jmse_loss.npz)."""
    n, k = maps.shape[:2]
    pred = maps.reshape(n, k, -1)
    gt = target.reshape(n, k, -1)
    loss = 0
    for j in range(k):
        if target_weight is not None:
            w = target_weight[:, j].reshape(n, 1)
            loss = loss + 0.5 * F.mse_loss(pred[:, j] * w, gt[:, j] * w, reduction='mean')
        else:
            loss = loss + 0.5 * F.mse_loss(pred[:, j], gt[:, j], reduction='mean')
    return loss / k


def composite_loss(out, target, joints_xy, img_size, w_hm=1.0, w_coor=0.1, w_cr=None, cr_indices=None,
                   target_cr=4.0 / 3.0, cr_loss_thres=0.15, cr_type='sl1', hm_type='mse', coor_type='l1'):
    """out = (maps [N,K,H,W], coords [N,K,2]); target [N,K,H,W]; joints_xy [N,K,2] in
    input-image pixels.  w_cr (with cr_indices) adds the cross-ratio term; hm_type / coor_type pick the
    criteria of the first two terms from loss_dict (function.py:17-20, 61-93)."""
    total = _composite_supervised(out, target, joints_xy, img_size, w_hm, w_coor, hm_type, coor_type)
    if w_cr is not None and isinstance(out, tuple):
        total = total + cross_ratio_loss(out[1], cr_indices, target_cr, cr_loss_thres, cr_type) * w_cr
    return total


def _composite_supervised(out, target, joints_xy, img_size, w_hm, w_coor, hm_type='mse', coor_type='l1'):
    maps, coords = out if isinstance(out, tuple) else (out, None)
    n, k = maps.shape[:2]
    pred = maps.reshape(n, k, -1)
    gt = target.reshape(n, k, -1)
    loss = 0
    for j in range(k):                      # function.py:103-111, joint by joint
        loss = loss + 0.5 * _CRIT[hm_type](pred[:, j], gt[:, j], reduction='mean')
    total = (loss / k) * w_hm
    if coords is None or not w_coor:
        return total
    cgt = joints_xy.clone().float()
    cgt[:, :, 0] /= img_size[0]
    cgt[:, :, 1] /= img_size[1]
    return total + _CRIT[coor_type](coords, cgt, reduction='mean') * w_coor


class HRNetTrainOracle(object):
    """state_dict (cloned) + Adam state; ``step`` returns (loss, maps, coords)."""

    def __init__(self, sd, cfgs, lr=1e-3, w_hm=1.0, w_coor=0.1, frozen_prefixes=(), cr=None, optim=None):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.cfgs = cfgs
        self.w = (w_hm, w_coor)
        self.cr = cr or {}       # composite_loss keywords: w_cr, cr_indices, target_cr, ...
        self.param_keys = [k for k in self.sd
                           if k.rsplit('.', 1)[-1] in ('weight', 'bias')
                           and not any(k.startswith(p) for p in frozen_prefixes)]
        for k in self.param_keys:
            self.sd[k].requires_grad_(True)
        # optim: keywords of prepare_optim (optimizer.py:8-40): optim_type 'adam' | 'sgd', momentum, weight_decay
        o = dict(optim or {})
        params = [self.sd[k] for k in self.param_keys]
        if o.get('optim_type', 'adam') == 'sgd':
            self.opt = torch.optim.SGD(params, lr=lr, momentum=o.get('momentum', 0.0),
                                       weight_decay=o.get('weight_decay', 0.0))
        else:
            self.opt = torch.optim.Adam(params, lr=lr, weight_decay=o.get('weight_decay', 0.0))

    def step(self, x, target, joints_xy, update=True, target_weight=None):
        self.opt.zero_grad()
        out = hrnet_oracle.hrnet_forward_train(self.sd, self.cfgs, x)
        if target_weight is not None:          # JointsMSELoss(use_target_weight=True): the heat-map head's criterion
            assert not isinstance(out, tuple)
            loss = joints_mse_loss(out, target, target_weight) * self.w[0]
        else:
            loss = composite_loss(out, target, joints_xy, self.cfgs['heatmapModel']['input_size'], *self.w, **self.cr)
        loss.backward()
        if update:
            self.opt.step()
        if not isinstance(out, tuple):
            return float(loss.detach()), out.detach(), None
        return float(loss.detach()), out[0].detach(), out[1].detach()

    def grads(self):
        return {k: self.sd[k].grad.clone() for k in self.param_keys}
