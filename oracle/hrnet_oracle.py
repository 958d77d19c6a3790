"""CPU oracle: HRNet heat-map / coordinate regression forward (inference).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

A *functional* restatement of the forward pass the reference builds out of
``torch.nn`` modules.  It walks a flat ``state_dict`` (the reference's
checkpoint layout, HC.pth) with the topology taken from the same config
dictionary, so it never instantiates a module tree.

Reference followed (all paths relative to /root/reference):
  * stem / layer1 / transitions / stages / heads:
    libs/model/heatmapModel/hrnet.py:563-614  (PoseHighResolutionNet.forward)
  * BasicBlock        hrnet.py:76-92     Bottleneck  hrnet.py:113-133
  * multi-scale fuse  hrnet.py:282-300 + layer construction :222-277
  * transition layers hrnet.py:471-510
  * coordinate head   hrnet.py:423-467, 601-608

Numerics: fp32, eval-mode BatchNorm with eps=1e-5 (torch default, never
overridden in the reference), nearest up-sampling with integer factors,
summation order of the fuse and residual adds kept as in the reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _conv(sd, key, x, stride=1, pad=0):
    return F.conv2d(x, sd[key + '.weight'], sd.get(key + '.bias'), stride, pad)


BN_MOMENTUM = 0.1        # hrnet.py:20
_TRAIN = [False]         # train-mode BatchNorm (batch statistics + running-stat update)


def _bn(sd, key, x):
    if _TRAIN[0]:
        sd[key + '.num_batches_tracked'] += 1
        return F.batch_norm(x, sd[key + '.running_mean'], sd[key + '.running_var'],
                            sd[key + '.weight'], sd[key + '.bias'],
                            True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(x, sd[key + '.running_mean'], sd[key + '.running_var'],
                        sd[key + '.weight'], sd[key + '.bias'],
                        False, 0.0, BN_EPS)


def _shortcut(sd, p, x, stride):
    """1x1(stride)+BN projection when the checkpoint has one (hrnet.py:29-42,
    :179-189, :515-521), identity otherwise."""
    if (p + '.downsample.0.weight') in sd:
        return _bn(sd, p + '.downsample.1',
                   _conv(sd, p + '.downsample.0', x, stride, 0))
    return x


def basic_block(sd, p, x, stride=1):
    """hrnet.py:76-92."""
    y = F.relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x, stride, 1)))
    y = _bn(sd, p + '.bn2', _conv(sd, p + '.conv2', y, 1, 1))
    y = y + _shortcut(sd, p, x, stride)
    return F.relu(y)


def bottleneck(sd, p, x, stride=1):
    """hrnet.py:113-133."""
    y = F.relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x, 1, 0)))
    y = F.relu(_bn(sd, p + '.bn2', _conv(sd, p + '.conv2', y, stride, 1)))
    y = _bn(sd, p + '.bn3', _conv(sd, p + '.conv3', y, 1, 0))
    y = y + _shortcut(sd, p, x, stride)
    return F.relu(y)


_BLOCK_FN = {'basic': basic_block, 'bottleneck': bottleneck}
_EXPANSION = {'basic': 1, 'bottleneck': 4}


def hr_module(sd, p, xs, stage_cfg, multi_scale_output):
    """One HighResolutionModule, hrnet.py:282-300."""
    nb = stage_cfg['num_branches']
    blk = _BLOCK_FN[stage_cfg['block']]
    xs = list(xs)
    for b in range(nb):
        for k in range(stage_cfg['num_blocks'][b]):
            xs[b] = blk(sd, '%s.branches.%d.%d' % (p, b, k), xs[b])
    if nb == 1:
        return xs
    outs = []
    for i in range(nb if multi_scale_output else 1):
        y = None
        for j in range(nb):
            q = '%s.fuse_layers.%d.%d' % (p, i, j)
            if j == i:
                t = xs[j]
            elif j > i:
                # 1x1 conv + BN + nearest upsample 2^(j-i)   (hrnet.py:232-243)
                t = _bn(sd, q + '.1', _conv(sd, q + '.0', xs[j]))
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:
                # (i-j) strided 3x3 convs, ReLU on all but the last (:246-274)
                t = xs[j]
                for k in range(i - j):
                    t = _bn(sd, '%s.%d.1' % (q, k),
                            _conv(sd, '%s.%d.0' % (q, k), t, 2, 1))
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def _transition(sd, name, prev, pre_ch, cur_ch):
    """hrnet.py:471-510 (construction) + :572-593 (use)."""
    out = []
    for i in range(len(cur_ch)):
        p = '%s.%d' % (name, i)
        if i < len(pre_ch):
            if cur_ch[i] != pre_ch[i]:
                # the reference feeds every non-identity transition from the
                # LAST tensor of the previous stage (hrnet.py:575,583,591)
                out.append(F.relu(_bn(sd, p + '.1', _conv(sd, p + '.0', prev[-1], 1, 1))))
            else:
                out.append(prev[i])
        else:
            t = prev[-1]
            for j in range(i + 1 - len(pre_ch)):
                q = '%s.%d' % (p, j)
                t = F.relu(_bn(sd, q + '.1', _conv(sd, q + '.0', t, 2, 1)))
            out.append(t)
    return out


def coordinate_ramps(map_w, map_h):
    """[1,2,H,W] fp32 x/y ramps linspace(0,1) incl. end points, hrnet.py:461-467."""
    xs = np.tile(np.linspace(0, 1, map_w), (map_h, 1))
    ys = np.tile(np.linspace(0, 1, map_h).reshape(map_h, 1), (1, map_w))
    return torch.from_numpy(np.stack([xs, ys])[None].astype(np.float32))


def hrnet_forward_train(sd, cfgs, x):
    """Train-mode forward (model.train(): BatchNorm on batch statistics, running
    statistics updated in ``sd``), differentiable w.r.t. the tensors of ``sd``
    that require grad.  trainer.py:191 ``prediction = model(data)``."""
    _TRAIN[0] = True
    try:
        return _hrnet_forward(sd, cfgs, x)
    finally:
        _TRAIN[0] = False


@torch.no_grad()
def hrnet_forward(sd, cfgs, x, return_trunk=False):
    return _hrnet_forward(sd, cfgs, x, return_trunk)


def _hrnet_forward(sd, cfgs, x, return_trunk=False):
    """Forward of PoseHighResolutionNet in eval mode.

    sd    flat state_dict (HC.pth layout), fp32 CPU tensors
    cfgs  the reference's config dict (needs cfgs['heatmapModel'])
    x     [N,C,H,W] fp32
    returns what hrnet.py:596-614 returns for the configured head type.
    """
    hm = cfgs['heatmapModel']
    extra = hm['extra']
    x = F.relu(_bn(sd, 'bn1', _conv(sd, 'conv1', x, 2, 1)))
    x = F.relu(_bn(sd, 'bn2', _conv(sd, 'conv2', x, 2, 1)))
    for k in range(4):
        x = bottleneck(sd, 'layer1.%d' % k, x)

    pre_ch = [256]
    ys = [x]
    for sname, tname in (('stage2', 'transition1'), ('stage3', 'transition2'),
                         ('stage4', 'transition3')):
        sc = extra[sname]
        cur_ch = [c * _EXPANSION[sc['block']] for c in sc['num_channels']]
        xs = _transition(sd, tname, ys, pre_ch, cur_ch)
        last_stage = sname == 'stage4'
        for m in range(sc['num_modules']):
            mso = not (last_stage and m == sc['num_modules'] - 1)
            xs = hr_module(sd, '%s.%d' % (sname, m), xs, sc, mso)
        ys = xs
        pre_ch = cur_ch
    trunk = ys[0]
    if return_trunk:
        return trunk

    head = hm['head_type']
    if head == 'heatmap':
        k = extra['final_conv_kernel']
        out = _conv(sd, 'final_layer', trunk, 1, 1 if k == 3 else 0)
        if hm.get('pixel_shuffle'):
            up = int(hm['heatmap_size'][0] / hm['input_size'][0] * 4)
            out = F.relu(_bn(sd, 'upsample_layer.1', _conv(sd, 'upsample_layer.0', out)))
            out = F.pixel_shuffle(out, up)
        return out
    if head == 'coordinates':
        maps = _conv(sd, 'head1.0', trunk)
        mw, mh = hm['heatmap_size']
        ramps = coordinate_ramps(mw, mh).expand(len(maps), -1, -1, -1)
        t = torch.cat([maps, ramps], dim=1)
        for k in range(4):
            t = basic_block(sd, 'head2.%d' % k, t, stride=2)
        t = torch.sigmoid(_conv(sd, 'head2.4', t))
        return maps, t.reshape(len(maps), -1, 2)
    raise NotImplementedError(head)
