"""CPU oracle: key-point decode from heat-maps (TEST INFRASTRUCTURE ONLY).

Reference followed (paths relative to /root/reference):
  * hard arg-max          libs/common/img_proc.py:608-637  get_max_preds
  * soft-arg-max (torch)  libs/common/img_proc.py:678-707  soft_arg_max
  * soft-arg-max (numpy)  libs/common/img_proc.py:639-676  soft_arg_max_np
  * coordinate-head decode libs/model/egonet.py:436-438   (coords *= resolution)

All functions take/return numpy arrays; fp32 arithmetic like the reference.
The arg-max *index* (int) is the bit-exact parity item.
"""
import numpy as np


def argmax_index(hm):
    """Flat arg-max index per map (first max on ties, like np.argmax).

    hm [N,K,H,W] fp32 -> idx [N,K] int64, maxvals [N,K,1] fp32
    (img_proc.py:618-624).
    """
    n, k = hm.shape[:2]
    flat = hm.reshape(n, k, -1)
    return np.argmax(flat, axis=2), np.amax(flat, axis=2).reshape(n, k, 1)


def get_max_preds(hm):
    """img_proc.py:608-637: (idx % W, floor(idx / W)) as fp32, zeroed where
    the maximum is not > 0."""
    w = hm.shape[3]
    idx, maxvals = argmax_index(hm)
    preds = np.empty(idx.shape + (2,), dtype=np.float32)
    preds[..., 0] = (idx % w).astype(np.float32)
    preds[..., 1] = np.floor(idx.astype(np.float32) / w)
    preds *= (maxvals > 0.0).astype(np.float32)
    return preds, maxvals


def soft_arg_max(hm):
    """img_proc.py:678-707 restated for CPU: softmax over the flattened H*W
    map (no temperature), marginal sums, index-weighted sums.

    Returns preds [N,K,2] (x,y) in heat-map pixels and maxvals [N,K,1] = raw
    (pre-softmax) maximum.  No visibility mask (the reference has none here).
    """
    n, k, h, w = hm.shape
    flat = hm.reshape(n, k, -1).astype(np.float32)
    maxvals = flat.max(axis=2).reshape(n, k, 1)
    e = np.exp(flat - flat.max(axis=2, keepdims=True))
    p = (e / e.sum(axis=2, keepdims=True)).astype(np.float32).reshape(n, k, h, w)
    px = p.sum(axis=2)                      # [N,K,W]
    py = p.sum(axis=3)                      # [N,K,H]
    x = (px * np.arange(w, dtype=np.float32)).sum(axis=2, keepdims=True)
    y = (py * np.arange(h, dtype=np.float32)).sum(axis=2, keepdims=True)
    return np.concatenate([x, y], axis=2).astype(np.float32), maxvals


def soft_arg_max_np(hm):
    """img_proc.py:639-676: normalise by the plain sum (the clip at :656 acts
    on a temporary and has no effect on the view that is used), marginals,
    index-weighted sums, masked by max > 0.  Does NOT mutate its input
    (the reference does; callers must not rely on that)."""
    n, k, h, w = hm.shape
    flat = hm.reshape(n, k, -1).astype(np.float32).copy()
    maxvals = flat.max(axis=2).reshape(n, k, 1)
    flat /= flat.sum(axis=2, keepdims=True)
    p = flat.reshape(n, k, h, w)
    x = (p.sum(axis=2) * np.arange(w, dtype=np.float32)).sum(axis=2, keepdims=True)
    y = (p.sum(axis=3) * np.arange(h, dtype=np.float32)).sum(axis=2, keepdims=True)
    preds = np.concatenate([x, y], axis=2)
    preds *= (maxvals > 0.0).astype(np.float32)
    return preds, maxvals


def coords_head_to_pixels(coords, resolution):
    """egonet.py:436-438: normalised (0,1) coordinates -> crop pixels,
    resolution = [width, height]."""
    return coords * np.asarray(resolution, dtype=coords.dtype).reshape(1, 1, 2)
