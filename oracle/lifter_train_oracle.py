"""CPU oracle: one training iteration of the FC lifter (TEST INFRASTRUCTURE ONLY).

Restates the reference hot loop for ``tools/train_lifting.py``:
  libs/trainer/trainer.py:183-209   zero_grad -> model(data) -> loss -> backward -> optim.step
  libs/model/FCmodel.py:33-43,92-105 train-mode forward (BatchNorm1d on batch
                                     statistics, momentum 0.1, Dropout)
  libs/loss/function.py:204-215     MSELoss1D(reduction='mean')
  libs/optimizer/optimizer.py:8-40  torch.optim.Adam(lr, weight_decay=0)
on a flat state_dict with torch autograd (functional ops, no nn.Module).
Dropout is the identity here (p = 0): the reference's mask stream is not
reproducible outside its own process, parity runs use p = 0.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

PARAM_LEAVES = ('weight', 'bias')


def _unit(sd, fc, bn, x, leaky=False):
    z = F.linear(x, sd[fc + '.weight'], sd[fc + '.bias'])
    z = F.batch_norm(z, sd[bn + '.running_mean'], sd[bn + '.running_var'], sd[bn + '.weight'], sd[bn + '.bias'],
                     True, BN_MOMENTUM, BN_EPS)
    sd[bn + '.num_batches_tracked'] += 1
    return F.leaky_relu(z) if leaky else F.relu(z)      # nn.LeakyReLU() / nn.ReLU, FCmodel.py:19-22


def forward_train(sd, x, num_blocks=2, leaky=False):
    y = _unit(sd, 'w1', 'batch_norm1', x, leaky)
    for b in range(num_blocks):
        p = 'res_blocks.%d' % b
        z = _unit(sd, p + '.w1', p + '.batch_norm1', y, leaky)
        z = _unit(sd, p + '.w2', p + '.batch_norm2', z, leaky)
        y = y + z
    return F.linear(y, sd['w2.weight'], sd['w2.bias'])


class LifterTrainOracle(object):
    """Holds a state_dict (cloned) + Adam state; ``step(x, target)`` returns the loss."""

    def __init__(self, sd, lr=1e-3, num_blocks=2, leaky=False, optim=None):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.num_blocks = num_blocks
        self.leaky = leaky
        self.param_keys = [k for k in self.sd if k.rsplit('.', 1)[-1] in PARAM_LEAVES]
        for k in self.param_keys:
            self.sd[k].requires_grad_(True)
        o = dict(optim or {})                 # optimizer.py:8-40: optim_type 'adam' | 'sgd', momentum, weight_decay
        params = [self.sd[k] for k in self.param_keys]
        if o.get('optim_type', 'adam') == 'sgd':
            self.opt = torch.optim.SGD(params, lr=lr, momentum=o.get('momentum', 0.0),
                                       weight_decay=o.get('weight_decay', 0.0))
        else:
            self.opt = torch.optim.Adam(params, lr=lr, weight_decay=o.get('weight_decay', 0.0))

    def step(self, x, target):
        self.opt.zero_grad()
        loss = F.mse_loss(forward_train(self.sd, x, self.num_blocks, self.leaky), target, reduction='mean')
        loss.backward()
        self.opt.step()
        return float(loss.detach())

    def grads(self):
        return {k: self.sd[k].grad.clone() for k in self.param_keys}
