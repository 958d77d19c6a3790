"""CPU oracle: FC lifter forward (TEST INFRASTRUCTURE ONLY).

Reference followed (paths relative to /root/reference):
  * FCModel.forward / get_representation   libs/model/FCmodel.py:92-105
  * ResidualBlock.forward                  libs/model/FCmodel.py:33-43
  * EgoNet.lift_2d_to_3d                   libs/model/egonet.py:469-486

Eval mode only (Dropout = identity, BatchNorm1d on running stats, eps 1e-5).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _lin(sd, k, x):
    return F.linear(x, sd[k + '.weight'], sd[k + '.bias'])


def _bn(sd, k, x):
    return F.batch_norm(x, sd[k + '.running_mean'], sd[k + '.running_var'],
                        sd[k + '.weight'], sd[k + '.bias'], False, 0.0, BN_EPS)


@torch.no_grad()
def lifter_forward(sd, x, num_blocks=2, leaky=False):
    """sd: L.pth layout; x [N,in] fp32 -> [N,out] fp32."""
    act = (lambda t: F.leaky_relu(t, 0.01)) if leaky else F.relu
    y = act(_bn(sd, 'batch_norm1', _lin(sd, 'w1', x)))
    for b in range(num_blocks):
        p = 'res_blocks.%d' % b
        z = act(_bn(sd, p + '.batch_norm1', _lin(sd, p + '.w1', y)))
        z = act(_bn(sd, p + '.batch_norm2', _lin(sd, p + '.w2', z)))
        y = y + z
    return _lin(sd, 'w2', y)


def lift_2d_to_3d(sd, stats, kpts_2d, num_blocks=2, leaky=False):
    """egonet.py:469-486 for one image: kpts_2d [n,66] float64 screen
    coordinates -> [n,32,3] float64."""
    data = ((kpts_2d - stats['mean_in']) / stats['std_in']).astype(np.float32)
    pred = lifter_forward(sd, torch.from_numpy(data), num_blocks, leaky).numpy()
    pred = pred * stats['std_out'] + stats['mean_out']
    return pred.reshape(len(pred), -1, 3)
