"""CPU oracle: one training iteration of the HRNet heat-map/coordinate model
(TEST INFRASTRUCTURE ONLY).

Restates, on a flat state_dict with torch autograd (functional ops only):
  libs/trainer/trainer.py:183-209    zero_grad -> model(data) -> loss -> backward -> optim.step
  libs/model/heatmapModel/hrnet.py   train-mode forward (oracle/hrnet_oracle.py, BatchNorm
                                     on batch statistics, momentum 0.1)
  libs/loss/function.py:61-202       JointsCompositeLoss with spec ['mse','l1',None]:
        L = w_hm * (1/K) sum_k 0.5 * MSE_mean(hm_k, tgt_k)            (:95-111)
          + w_coor * L1_mean(coords_pred, joints_xy / img_size)       (:155-168, :186-199)
     (KITTI_train_IGRs.yml:88-89: weights 1.0 / 0.1, cross-ratio term off)
  libs/optimizer/optimizer.py:8-40   Adam(lr, weight_decay 0)
"""
import torch
import torch.nn.functional as F

from . import hrnet_oracle


def composite_loss(out, target, joints_xy, img_size, w_hm=1.0, w_coor=0.1):
    """out = (maps [N,K,H,W], coords [N,K,2]); target [N,K,H,W]; joints_xy [N,K,2] in
    input-image pixels."""
    maps, coords = out if isinstance(out, tuple) else (out, None)
    n, k = maps.shape[:2]
    pred = maps.reshape(n, k, -1)
    gt = target.reshape(n, k, -1)
    loss = 0
    for j in range(k):                      # function.py:103-111, joint by joint
        loss = loss + 0.5 * F.mse_loss(pred[:, j], gt[:, j], reduction='mean')
    total = (loss / k) * w_hm
    if coords is None or not w_coor:
        return total
    cgt = joints_xy.clone().float()
    cgt[:, :, 0] /= img_size[0]
    cgt[:, :, 1] /= img_size[1]
    return total + F.l1_loss(coords, cgt, reduction='mean') * w_coor


class HRNetTrainOracle(object):
    """state_dict (cloned) + Adam state; ``step`` returns (loss, maps, coords)."""

    def __init__(self, sd, cfgs, lr=1e-3, w_hm=1.0, w_coor=0.1, frozen_prefixes=()):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.cfgs = cfgs
        self.w = (w_hm, w_coor)
        self.param_keys = [k for k in self.sd
                           if k.rsplit('.', 1)[-1] in ('weight', 'bias')
                           and not any(k.startswith(p) for p in frozen_prefixes)]
        for k in self.param_keys:
            self.sd[k].requires_grad_(True)
        self.opt = torch.optim.Adam([self.sd[k] for k in self.param_keys], lr=lr)

    def step(self, x, target, joints_xy, update=True):
        self.opt.zero_grad()
        out = hrnet_oracle.hrnet_forward_train(self.sd, self.cfgs, x)
        loss = composite_loss(out, target, joints_xy, self.cfgs['heatmapModel']['input_size'], *self.w)
        loss.backward()
        if update:
            self.opt.step()
        if not isinstance(out, tuple):
            return float(loss.detach()), out.detach(), None
        return float(loss.detach()), out[0].detach(), out[1].detach()

    def grads(self):
        return {k: self.sd[k].grad.clone() for k in self.param_keys}
