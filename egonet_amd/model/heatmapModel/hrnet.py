"""HRNet heat-map / coordinate regression model -- MI355X build.

Drop-in for the reference's ``libs/model/heatmapModel/hrnet.py``: same factory
(``get_pose_net(cfgs, is_train)``, reference hrnet.py:675-690), same config
keys, same ``forward`` contract per head type (hrnet.py:596-614) and the same
``state_dict`` layout (1 828 entries for W48 + coordinates head), so ``HC.pth``
checkpoints load unchanged and ``tools/inference.py`` / ``EgoNet`` can use it.

What differs is execution.  The module tree only *owns parameters*; for CUDA
inputs in eval mode ``forward`` hands the batch to ``egonet_amd.engine``, which
runs the whole network as hand-written gfx950 HIP kernels (fused
conv+BN(+residual)+ReLU on fp32 MFMA, NHWC activations, fused multi-resolution
sums).  CPU tensors (``get_model_summary``'s warm-up forward at
tools/train_IGRs.py:54-57, BASELINE config 1) and training-mode forwards run
the same graph through ``torch.nn.functional`` on the tensor's own device.
"""
import logging
import os

import numpy as np
import torch
import torch.nn as nn

logger = logging.getLogger(__name__)
_MOMENTUM = 0.1



# Every registration of a submodule anywhere (add_module / __setattr__ / container item assignment) bumps this counter:
# the hook cache below is valid for the module TREE it was collected from (a block replaced or wrapped after
# .eval() and then given a hook was not seen).  torch's global registration hook costs nothing on the forward path.
_TREE_GENERATION = [0]


def _bump_tree_generation(module, name, submodule):
    _TREE_GENERATION[0] += 1


_REG_HOOK = getattr(nn.modules.module, 'register_module_module_registration_hook', None)
if _REG_HOOK is not None:
    _REG_HOOK(_bump_tree_generation)


def _has_submodule_hooks(module):
    """True if any SUBmodule carries a forward / backward hook (hooks on the module itself fire around its forward
    whichever path runs inside).  Called on every train-mode forward of a launch-bound step: the hook dictionaries of
    the ~1.5 k submodules (register_*_hook mutates them in place) are collected once per module tree -- the cache is
    dropped by ``.train()`` / ``.eval()`` and whenever ANY submodule is registered anywhere (_TREE_GENERATION) -- 0.015 ms
    per check instead of a 0.9 ms Python walk.  The cache lives outside ``__dict__`` pickling
    (``__getstate__`` drops it)."""
    cache = module.__dict__.get('_hook_dicts')
    if cache is None or cache[0] != _TREE_GENERATION[0] or _REG_HOOK is None:
        dicts = [d for m in module.modules() if m is not module
                 for d in (m._forward_hooks, m._forward_pre_hooks, m._backward_hooks,
                           getattr(m, '_backward_pre_hooks', None)) if d is not None]
        cache = (_TREE_GENERATION[0], dicts)
        module.__dict__['_hook_dicts'] = cache
    return any(cache[1])


def _conv(cin, cout, k, stride=1, bias=False):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k - 1) // 2, bias=bias)


def _unit(cin, cout, k, stride, relu, momentum=None):
    """conv + BN (+ ReLU) as a Sequential -> keys ``.0.weight``, ``.1.*``.
    BatchNorm momentum: the reference passes 0.1 explicitly in blocks and leaves
    the default (also 0.1) elsewhere."""
    mods = [_conv(cin, cout, k, stride), nn.BatchNorm2d(cout, momentum=momentum or _MOMENTUM)]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


class _Residual(nn.Module):
    """Residual block driven by a kernel-size table.

    'basic'      : 3x3 -> 3x3                 (reference BasicBlock, hrnet.py:63-92)
    'bottleneck' : 1x1 -> 3x3 -> 1x1 (x4)     (reference Bottleneck, hrnet.py:95-133)
    The stride sits on the first 3x3; parameters are registered as
    conv1/bn1 ... convK/bnK (+ ``downsample``) to keep the checkpoint keys.
    """
    KERNELS = {'basic': (3, 3), 'bottleneck': (1, 3, 1)}
    EXPANSION = {'basic': 1, 'bottleneck': 4}

    def __init__(self, kind, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.kind = kind
        ks = self.KERNELS[kind]
        widths = [planes] * (len(ks) - 1) + [planes * self.EXPANSION[kind]]
        first3 = ks.index(3)
        c = cin
        for i, (k, w) in enumerate(zip(ks, widths), start=1):
            setattr(self, 'conv%d' % i, _conv(c, w, k, stride if i - 1 == first3 else 1))
            setattr(self, 'bn%d' % i, nn.BatchNorm2d(w, momentum=_MOMENTUM))
            c = w
        self.depth = len(ks)
        self.stride = stride
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        y = x
        for i in range(1, self.depth + 1):
            y = getattr(self, 'bn%d' % i)(getattr(self, 'conv%d' % i)(y))
            if i < self.depth:
                y = self.relu(y)
        y += x if self.downsample is None else self.downsample(x)
        return self.relu(y)


def BasicBlock(cin, planes, stride=1, downsample=None):
    return _Residual('basic', cin, planes, stride, downsample)


def Bottleneck(cin, planes, stride=1, downsample=None):
    return _Residual('bottleneck', cin, planes, stride, downsample)


def _projection(cin, cout, stride):
    """1x1(stride) conv + BN shortcut (hrnet.py:29-42, 179-189, 515-521)."""
    return nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout, momentum=_MOMENTUM))


def _stack(kind, cin, planes, n, stride=1):
    cout = planes * _Residual.EXPANSION[kind]
    proj = _projection(cin, cout, stride) if (stride != 1 or cin != cout) else None
    blocks = [_Residual(kind, cin, planes, stride, proj)]
    blocks += [_Residual(kind, cout, planes) for _ in range(n - 1)]
    return nn.Sequential(*blocks), cout


class HighResolutionModule(nn.Module):
    """Parallel branches + multi-resolution exchange (hrnet.py:136-300)."""

    def __init__(self, kind, num_blocks, in_channels, channels, multi_scale_output=True):
        super().__init__()
        nb = len(channels)
        if not (nb == len(num_blocks) == len(in_channels)):
            raise ValueError('branches/blocks/channels length mismatch: %d %d %d'
                             % (nb, len(num_blocks), len(in_channels)))
        self.num_branches = nb
        self.multi_scale_output = multi_scale_output
        branches, widths = [], []
        for b in range(nb):
            seq, w = _stack(kind, in_channels[b], channels[b], num_blocks[b])
            branches.append(seq)
            widths.append(w)
        self.branches = nn.ModuleList(branches)
        self.out_channels = widths
        self.fuse_layers = self._exchange(widths) if nb > 1 else None
        self.relu = nn.ReLU(True)

    def _exchange(self, w):
        nb = self.num_branches
        rows = []
        for i in range(nb if self.multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j == i:
                    row.append(None)
                elif j > i:       # coarser -> finer: 1x1 + BN + nearest upsample
                    row.append(nn.Sequential(_conv(w[j], w[i], 1), nn.BatchNorm2d(w[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                else:             # finer -> coarser: (i-j) strided 3x3, ReLU between
                    steps = i - j
                    row.append(nn.Sequential(*[
                        _unit(w[j], w[i] if s == steps - 1 else w[j], 3, 2, relu=(s != steps - 1))
                        for s in range(steps)]))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def forward(self, xs):
        xs = [br(x) for br, x in zip(self.branches, xs)]
        if self.fuse_layers is None:
            return xs
        outs = []
        for i, row in enumerate(self.fuse_layers):
            y = None
            for j in range(self.num_branches):
                t = xs[j] if row[j] is None else row[j](xs[j])
                y = t if y is None else y + t
            outs.append(self.relu(y))
        return outs


class PoseHighResolutionNet(nn.Module):
    """Reference: hrnet.py:309-667."""

    def __init__(self, cfgs, **kwargs):
        super().__init__()
        hm = cfgs['heatmapModel']
        extra = hm['extra']
        self.num_joints = hm['num_joints']
        self.head_type = hm['head_type']
        self.pixel_shuffle = hm['pixel_shuffle']
        self.pretrained_layers = extra['pretrained_layers']
        self.stage_cfgs = [extra['stage2'], extra['stage3'], extra['stage4']]
        self.stage2_cfg, self.stage3_cfg, self.stage4_cfg = self.stage_cfgs

        self.conv1 = _conv(3, 64, 3, 2)
        self.bn1 = nn.BatchNorm2d(64, momentum=_MOMENTUM)
        self.conv2 = _conv(64, 64, 3, 2)
        self.bn2 = nn.BatchNorm2d(64, momentum=_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.layer1, width = _stack('bottleneck', 64, 64, 4)

        pre = [width]
        for idx, sc in enumerate(self.stage_cfgs, start=1):
            kind = sc['block']
            cur = [c * _Residual.EXPANSION[kind] for c in sc['num_channels']]
            setattr(self, 'transition%d' % idx, self._transition(pre, cur))
            mods = []
            for m in range(sc['num_modules']):
                last = (idx == 3 and m == sc['num_modules'] - 1)
                mod = HighResolutionModule(kind, sc['num_blocks'], cur,
                                           sc['num_channels'], multi_scale_output=not last)
                mods.append(mod)
                cur = mod.out_channels
            setattr(self, 'stage%d' % (idx + 1), nn.Sequential(*mods))
            pre = cur
        trunk_c = pre[0]

        if self.head_type == 'heatmap':
            k = extra['final_conv_kernel']
            self.final_layer = _conv(trunk_c, self.num_joints, k, 1, bias=True)
            if self.pixel_shuffle:
                self.upsamp_fact = int(hm['heatmap_size'][0] / hm['input_size'][0] * 4)
                up2 = self.upsamp_fact ** 2
                self.upsample_layer = nn.Sequential(
                    nn.Conv2d(self.num_joints, self.num_joints * up2, kernel_size=1),
                    nn.BatchNorm2d(self.num_joints * up2), nn.ReLU(inplace=True),
                    nn.PixelShuffle(self.upsamp_fact))
        elif self.head_type == 'coordinates':
            j = self.num_joints
            map_w, map_h = hm['heatmap_size']
            self.head1 = nn.Sequential(_conv(trunk_c, j, 1, 1, bias=True))
            chain, c = [], j + 2
            for _ in range(4):
                chain.append(_Residual('basic', c, 2 * j, 2, _projection(c, 2 * j, 2)))
                c = 2 * j
            chain += [nn.Conv2d(2 * j, 2 * j, kernel_size=(int(map_h / 16), int(map_w / 16))),
                      nn.Sigmoid()]
            self.head2 = nn.Sequential(*chain)
            # plain attribute like the reference (not a buffer -> not in checkpoints)
            gx = np.tile(np.linspace(0, 1, map_w), (map_h, 1))
            gy = np.tile(np.linspace(0, 1, map_h).reshape(map_h, 1), (1, map_w))
            self.coor_maps = torch.from_numpy(np.stack([gx, gy])[None].astype(np.float32))
        elif self.head_type == 'angleregression':
            c = 256
            self.head = nn.Sequential(
                nn.Conv2d(trunk_c, c, kernel_size=1),
                *[_Residual('basic', c, c, 2, _projection(c, c, 2)) for _ in range(4)],
                nn.AvgPool2d(kernel_size=4))
            self.final_fc = nn.Sequential(nn.Linear(256, 256), nn.BatchNorm1d(256),
                                          nn.ReLU(inplace=True), nn.Linear(256, 2))
        else:
            raise NotImplementedError
        self._engine = None

    @staticmethod
    def _transition(pre, cur):
        """hrnet.py:471-510: 3x3 s1 where the width changes, a chain of 3x3 s2
        for every new (coarser) branch, None where nothing changes."""
        layers = []
        for i, c in enumerate(cur):
            if i < len(pre):
                layers.append(_unit(pre[i], c, 3, 1, relu=True) if c != pre[i] else None)
            else:
                steps = i + 1 - len(pre)
                layers.append(nn.Sequential(*[
                    _unit(pre[-1], c if s == steps - 1 else pre[-1], 3, 2, relu=True)
                    for s in range(steps)]))
        return nn.ModuleList(layers)

    # -- execution ---------------------------------------------------------
    def forward(self, x):
        """CUDA input: the native HIP kernels, in both modes.

        Eval mode: the recorded inference program, whatever the autograd mode -- the reference's validation
        loop calls ``model.eval(); model(data)`` with grad enabled (libs/trainer/trainer.py:421) and
        ``EgoNet.get_keypoints`` has no ``no_grad`` of its own (libs/model/egonet.py:434).  The outputs are
        plain tensors (no autograd graph); a caller that wants to differentiate an eval-mode forward gives
        an input with ``requires_grad`` (saliency) or sets ``model.hip_eval = False`` (frozen-BatchNorm
        fine-tuning through the module graph) and gets the torch graph.
        Train mode under autograd -- the reference's hot loop ``model(data); loss.backward(); optim.step()``
        (libs/trainer/trainer.py:183-209), unchanged: ONE autograd node whose forward / backward are the
        native train-mode tape (egonet_amd.autograd.HRNetAutograd); torch owns loss and optimiser.
        ``EGONET_AMD_AUTOGRAD=0``, an input that requires a gradient, heads the tape does not train
        (pixel shuffle / angle regression) and train mode WITHOUT autograd (``get_model_summary``) run the
        module graph in torch.  CPU tensors: torch on the CPU (the reference's CPU path)."""
        if x.is_cuda:
            wants_input_grad = torch.is_grad_enabled() and x.requires_grad
            if not self.training:
                if self.hip_eval and not wants_input_grad:
                    return self._hip_engine().forward(x)
            elif torch.is_grad_enabled() and not wants_input_grad and self._native_autograd_ok():
                return self._autograd_bridge()(x)
        return self._torch_forward(x)

    hip_eval = True          # eval-mode CUDA forwards run the HIP program (see forward)

    def _native_autograd_ok(self):
        """The train-mode bridge runs ONE autograd node on the native tape: the nn.Module graph below this module is
        not executed.  Whatever needs that graph takes the module's torch forward instead (explicit opt-outs, never a
        silent difference): forward / backward hooks on any submodule (they would not fire), nn.DataParallel replicas
        (a replica is re-created on every forward, so its bridge -- walker, packed filters, side stream -- would be
        rebuilt per step; the rank-per-GPU launcher is the performance path), head variants the tape does not cover."""
        return (os.environ.get('EGONET_AMD_AUTOGRAD', '1') != '0' and not self.pixel_shuffle
                and self.head_type in ('coordinates', 'heatmap')
                and not getattr(self, '_is_replica', False)
                and not _has_submodule_hooks(self)
                and any(p.requires_grad for p in self.parameters()))

    def _autograd_bridge(self):
        from egonet_amd import autograd
        b = self.__dict__.get('_bridge')
        if b is None or b.model is not self:        # (an nn.DataParallel replica builds its own, like _hip_engine)
            b = autograd.HRNetAutograd(self)
            self.__dict__['_bridge'] = b
        return b

    def __getstate__(self):
        # torch.save(model) / copy.deepcopy: the launch programs, the autograd bridge and the hook cache are
        # per-process device state (ctypes handles, streams) -- rebuilt on first use, never pickled
        state = dict(self.__dict__)
        state['_engine'] = None
        state.pop('_bridge', None)
        state.pop('_hook_dicts', None)
        return state

    def train(self, mode=True):
        if mode:                 # the weights are about to change: drop programs and packed blobs
            self._engine = None
        self.__dict__.pop('_hook_dicts', None)      # (submodules may have been replaced since: _has_submodule_hooks)
        return super().train(mode)

    def _hip_engine(self):
        from egonet_amd import engine
        # `is not self`: an nn.DataParallel replica is a shallow copy whose `_engine` attribute still
        # points at the original module's engine (and its packed weights) -- a replica folds its own
        # parameters (re-built on every forward there: DataParallel is supported for API
        # compatibility, the rank-per-GPU launcher is the performance path)
        if self._engine is None or self._engine.model is not self:
            self._engine = engine.HRNetEngine(self)
        return self._engine

    def _trunk(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.relu(self.bn2(self.conv2(x)))
        ys = [self.layer1(x)]
        for idx in (1, 2, 3):
            trans = getattr(self, 'transition%d' % idx)
            xs = [ys[i] if t is None else t(ys[-1]) for i, t in enumerate(trans)]
            ys = getattr(self, 'stage%d' % (idx + 1))(xs)
        return ys[0]

    def _torch_forward(self, x):
        y = self._trunk(x)
        if self.head_type == 'heatmap':
            y = self.final_layer(y)
            return self.upsample_layer(y) if self.pixel_shuffle else y
        if self.head_type == 'coordinates':
            maps = self.head1(y)
            ramps = self.coor_maps.to(maps.device).expand(len(maps), -1, -1, -1)
            coords = self.head2(torch.cat([maps, ramps], dim=1))
            return maps, coords.view(len(x), -1, 2)
        maps = self.head(y)
        return self.final_fc(maps.reshape(len(maps), -1))

    # -- checkpoint / init plumbing (hrnet.py:616-667) ----------------------
    def init_weights(self, pretrained=''):
        logger.info('=> init weights from normal distribution')
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if os.path.isfile(pretrained):
            state = torch.load(pretrained)
            logger.info('=> loading pretrained model {}'.format(pretrained))
            keep = {k: v for k, v in state.items()
                    if self.pretrained_layers[0] == '*' or k.split('.')[0] in self.pretrained_layers}
            self.load_state_dict(keep, strict=False)
            logger.info('{:d} modules initialized.'.format(len(keep)))
        elif pretrained:
            logger.error('=> please download pre-trained models first!')
            raise ValueError('{} does not exist!'.format(pretrained))

    def modify_input_channel(self, num_channels):
        if num_channels == 3:
            return
        wide = _conv(num_channels, 64, 3, 2)
        with torch.no_grad():
            wide.weight[:, :3] = self.conv1.weight
        self.conv1 = wide
        self._engine = None

    def load_my_state_dict(self, state_dict):
        own = self.state_dict()
        for name, value in state_dict.items():
            if name in own:
                own[name].copy_(value.data)

    def _apply(self, fn, *a, **k):       # .cuda()/.to()/.float(): packed weights are stale
        self._engine = None
        self.__dict__.pop('_bridge', None)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)


def is_freezed(name, freeze_names):
    return any(name.startswith(p) for p in freeze_names)


def get_pose_net(cfgs, is_train, **kwargs):
    """Factory with the reference's signature and side effects (hrnet.py:675-690)."""
    model = PoseHighResolutionNet(cfgs, **kwargs)
    hm = cfgs['heatmapModel']
    if is_train and hm['init_weights']:
        model.init_weights(hm.get('pretrained', ''))
    frozen = hm['extra'].get('freeze_layers', [])
    for name, param in model.named_parameters():
        if is_freezed(name, frozen):
            param.requires_grad = False
            print('{:s} freezed during training.'.format(name))
    if hm.get('add_xy'):
        model.modify_input_channel(5)
    return model
