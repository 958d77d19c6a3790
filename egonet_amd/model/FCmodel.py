"""Fully-connected 2D->3D lifter -- MI355X build.

Drop-in for the reference's ``libs/model/FCmodel.py``: ``get_fc_model(stage_id,
cfgs, input_size, output_size)`` (FCmodel.py:107-121) returns a module with the
reference's ``forward`` / ``get_representation`` contract and ``state_dict``
layout (37 entries: ``w1, batch_norm1, res_blocks.{i}.{w1,batch_norm1,w2,
batch_norm2}, w2``), so ``L.pth`` loads unchanged.

Eval-mode CUDA forwards run as six fused fp32-MFMA GEMM launches
(``egonet_amd.engine.LifterEngine``); CPU tensors and training-mode forwards
use ``torch.nn.functional`` on the tensor's own device.
"""
import torch
import torch.nn as nn


def _activation(leaky):
    return nn.LeakyReLU(inplace=True) if leaky else nn.ReLU(inplace=True)


class ResidualBlock(nn.Module):
    """x + drop(act(bn2(w2(drop(act(bn1(w1 x)))))))   (FCmodel.py:9-43)."""

    def __init__(self, num_neurons, p_dropout=0.5, kaiming=False, leaky=False):
        super().__init__()
        self.num_neurons, self.p_dropout, self.leaky = num_neurons, p_dropout, leaky
        self.relu = _activation(leaky)
        self.dropout = nn.Dropout(p_dropout)
        for i in (1, 2):
            fc = nn.Linear(num_neurons, num_neurons)
            if kaiming:
                nn.init.kaiming_normal_(fc.weight.data)
            setattr(self, 'w%d' % i, fc)
            setattr(self, 'batch_norm%d' % i, nn.BatchNorm1d(num_neurons))

    def forward(self, x):
        y = x
        for i in (1, 2):
            y = getattr(self, 'batch_norm%d' % i)(getattr(self, 'w%d' % i)(y))
            y = self.dropout(self.relu(y))
        return x + y


class FCModel(nn.Module):
    """FCmodel.py:45-105."""

    def __init__(self, stage_id=1, num_neurons=1024, num_blocks=2, p_dropout=0.5, norm_twoD=False,
                 kaiming=False, refine_3d=False, leaky=False, dm=False, input_size=32, output_size=64):
        super().__init__()
        self.stage_id, self.num_neurons, self.num_blocks = stage_id, num_neurons, num_blocks
        self.p_dropout, self.refine_3d, self.leaky, self.dm = p_dropout, refine_3d, leaky, dm
        self.input_size, self.output_size = input_size, output_size
        self.w1 = nn.Linear(input_size, num_neurons)
        self.batch_norm1 = nn.BatchNorm1d(num_neurons)
        self.res_blocks = nn.ModuleList(
            [ResidualBlock(num_neurons, p_dropout, leaky=leaky) for _ in range(num_blocks)])
        self.w2 = nn.Linear(num_neurons, output_size)
        self.relu = _activation(leaky)
        self.dropout = nn.Dropout(p_dropout)
        if kaiming:
            nn.init.kaiming_normal_(self.w1.weight.data)
            nn.init.kaiming_normal_(self.w2.weight.data)
        self._engine = None

    hip_eval = True          # eval-mode CUDA forwards run the HIP program

    def _hip_ok(self, x):
        # eval mode + CUDA input: the HIP program, whatever the autograd mode (the reference evaluates with
        # grad enabled, libs/trainer/trainer.py:421) -- unless the caller asks for a graph: an input that
        # requires a gradient, or ``model.hip_eval = False``
        return x.is_cuda and not self.training and self.hip_eval and not (torch.is_grad_enabled() and x.requires_grad) \
            and not self._dropout_active()

    def _dropout_active(self):
        """``model.eval()`` followed by ``Dropout.train()`` on the dropout layers -- the reference's
        ``testing_settings.apply_dropout`` (libs/trainer/trainer.py:424-428): BatchNorm on running statistics, dropout
        masks drawn.  The eval-mode HIP program has no dropout, so that state takes the module's torch graph (the
        reference's own path) instead of silently ignoring the masks [round 6].  The dropout layers are collected once
        per module tree (heatmapModel.hrnet._TREE_GENERATION)."""
        from .heatmapModel.hrnet import _TREE_GENERATION
        cache = self.__dict__.get('_dropouts')
        if cache is None or cache[0] != _TREE_GENERATION[0]:
            cache = (_TREE_GENERATION[0], [m for m in self.modules() if isinstance(m, nn.Dropout)])
            self.__dict__['_dropouts'] = cache
        return any(m.training and m.p > 0 for m in cache[1])

    def _native_autograd_ok(self, x):
        # train mode under autograd (the reference's hot loop, libs/trainer/trainer.py:183-209, unchanged):
        # one autograd node on the native kernels (egonet_amd.autograd.LifterAutograd)
        import os
        # (submodule hooks would not fire inside the single native node, DataParallel replicas would rebuild the
        # bridge on every forward: both take the module's torch graph -- see PoseHighResolutionNet._native_autograd_ok)
        from .heatmapModel.hrnet import _has_submodule_hooks
        return (x.is_cuda and self.training and torch.is_grad_enabled() and not x.requires_grad and x.shape[0] > 1
                and os.environ.get('EGONET_AMD_AUTOGRAD', '1') != '0'
                and not getattr(self, '_is_replica', False) and not _has_submodule_hooks(self)
                and all(p.requires_grad for p in self.parameters()))

    def _autograd_bridge(self):
        from egonet_amd import autograd
        b = self.__dict__.get('_bridge')
        if b is None or b.model is not self:
            b = autograd.LifterAutograd(self)
            self.__dict__['_bridge'] = b
        return b

    def __getstate__(self):
        # torch.save(model) / copy.deepcopy: the launch programs, the autograd bridge and the hook cache are
        # per-process device state (ctypes handles, streams) -- rebuilt on first use, never pickled
        state = dict(self.__dict__)
        state['_engine'] = None
        state.pop('_bridge', None)
        state.pop('_hook_dicts', None)
        state.pop('_dropouts', None)
        return state

    def train(self, mode=True):
        if mode:                 # the weights are about to change: drop programs and packed blobs
            self._engine = None
        self.__dict__.pop('_hook_dicts', None)      # (heatmapModel.hrnet._has_submodule_hooks' cache)
        return super().train(mode)

    def _hip_engine(self):
        from egonet_amd import engine
        # `is not self`: an nn.DataParallel replica is a shallow copy whose `_engine` attribute still
        # points at the original module's engine (and its packed weights) -- a replica folds its own
        # parameters (re-built on every forward there: DataParallel is supported for API
        # compatibility, the rank-per-GPU launcher is the performance path)
        if self._engine is None or self._engine.model is not self:
            self._engine = engine.LifterEngine(self)
        return self._engine

    def forward(self, x):
        if self._hip_ok(x):
            return self._hip_engine().forward(x)
        if self._native_autograd_ok(x):
            return self._autograd_bridge()(x)
        return self.w2(self.get_representation(x))

    def get_representation(self, x):
        y = self.dropout(self.relu(self.batch_norm1(self.w1(x))))
        for blk in self.res_blocks:
            y = blk(y)
        return y

    def _apply(self, fn, *a, **k):
        self._engine = None
        self.__dict__.pop('_bridge', None)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)


def get_fc_model(stage_id, cfgs, input_size, output_size, architecture_type='FCModel'):
    c = cfgs[architecture_type]
    return FCModel(stage_id=stage_id, refine_3d=c['refine_3d'], norm_twoD=c['norm_twoD'],
                   num_blocks=c['num_blocks'], input_size=input_size, output_size=output_size,
                   num_neurons=c['num_neurons'], p_dropout=c['dropout'], leaky=c['leaky'])


def get_cascade():
    return nn.ModuleList([])
