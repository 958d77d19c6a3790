"""Deterministic synthetic checkpoints and inputs.

The reference ships no weights (docs/preparation.md:28-29 points at a
Google-Drive folder), so benchmarks and parity tests run on *synthetic*
checkpoints.  Every tensor is drawn from its own generator seeded by
``crc32(key) ^ seed``; the values therefore depend only on the state_dict key
and shape, never on module construction order, and can be regenerated
bit-identically on the GPU box (torch's CPU Philox/MT generators are
platform independent for a given torch version).

The scales keep activations O(1) through ~70 conv+BN layers (SURVEY.md
section 8c-iii): conv/linear weights U(-1,1)/sqrt(fan_in), BN gamma in
[0.5,1.5], beta/running_mean ~ 0.2*N(0,1), running_var in [0.5,1.5].
"""
import zlib

import numpy as np
import torch


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def synth_tensor(key, shape, seed=0):
    shape = tuple(shape)
    g = _gen(key, seed)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long)
    if leaf == 'running_var':
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == 'running_mean':
        return 0.2 * torch.randn(shape, generator=g)
    if leaf == 'weight' and len(shape) == 1:          # BN gamma
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == 'bias':
        return 0.2 * torch.randn(shape, generator=g)
    if leaf == 'weight':
        fan_in = int(np.prod(shape[1:]))
        return (torch.rand(shape, generator=g) * 2 - 1) / np.sqrt(fan_in)
    raise KeyError('no synthetic recipe for ' + key)


def synth_state_dict(template, seed=0):
    """template: mapping key -> tensor (shapes are read, values ignored)."""
    out = {}
    for k, v in template.items():
        out[k] = synth_tensor(k, v.shape, seed).to(v.dtype if v.dtype != torch.long else torch.long)
    return out


def synth_crops(n, c=3, h=256, w=256, seed=0):
    g = torch.Generator()
    g.manual_seed(977 + seed)
    return torch.randn(n, c, h, w, generator=g)


def synth_lifter_stats(n_in=66, n_out=96, seed=0):
    """A plausible LS.npy dict (train_lifting.py:54): float64 [1,n] arrays."""
    rng = np.random.RandomState(4242 + seed)
    return {'mean_in': rng.uniform(300, 900, (1, n_in)),
            'std_in': rng.uniform(40, 120, (1, n_in)),
            'mean_out': rng.uniform(-1, 1, (1, n_out)),
            'std_out': rng.uniform(0.3, 1.5, (1, n_out))}


def synth_boxes(n, seed=0, img_w=1242, img_h=375):
    """n KITTI-like 2D boxes [x1,y1,x2,y2] (float64)."""
    rng = np.random.RandomState(99 + seed)
    w = rng.uniform(30, 320, n)
    h = w * rng.uniform(0.4, 1.3, n)
    x1 = rng.uniform(0, img_w - w)
    y1 = rng.uniform(0, np.maximum(img_h - h, 1))
    return np.stack([x1, y1, x1 + w, y1 + h], axis=1)
