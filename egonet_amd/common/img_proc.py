"""Key-point decode and crop geometry -- the hot-path subset of the reference's
``libs/common/img_proc.py`` with the same function names and return values.

Decode runs on the GPU (``csrc/decode.hip``, one wavefront per heat-map):
  ``soft_arg_max(t)``     reference img_proc.py:678-707 (CUDA tensor in/out)
  ``get_max_preds(a)``    reference img_proc.py:608-637 (numpy in/out like the
                          reference; CUDA tensors are accepted and stay on device)
  ``hard_arg_max(t)``     get_max_preds + the integer arg-max index
Host helpers (pure Python, per bounding box): ``modify_bbox`` & co.
(img_proc.py:411-459).
"""
import numpy as np
import torch

from .. import _lib

SIZE = 200.0


def get_cr_indices():
    """The 12 four-point lines of the 33-joint cuboid the cross-ratio term runs over
    (car_instance.py:83-119): joint 0 = centre, 1..8 = corners, 9+l / 21+l = the two
    interpolated points on edge l, whose end corners are (parent[l], child[l]) -- edges
    0-3 along h, 4-7 along l, 8-11 along w.  Row l = [parent, 9+l, 21+l, child]."""
    parent = [1, 3, 5, 7, 1, 2, 3, 4, 1, 2, 5, 6]
    child = [2, 4, 6, 8, 5, 6, 7, 8, 3, 4, 7, 8]
    return np.array([[parent[e], 9 + e, 21 + e, child[e]] for e in range(12)], dtype=np.int64)


CR_INDICES_BBOX12 = get_cr_indices()


def _decode(t, mode, want_idx=True):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise TypeError('decode kernels need a CUDA tensor; there is no CPU fallback')
    assert t.dim() == 4, 'batch_images should be 4-ndim'
    t = t.contiguous().float()
    n, k, h, w = t.shape
    xy = torch.empty(n, k, 2, dtype=torch.float32, device=t.device)
    mx = torch.empty(n, k, 1, dtype=torch.float32, device=t.device)
    idx = torch.empty(n, k, dtype=torch.int32, device=t.device) if want_idx else None
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().egn_decode_heatmaps_f32(
            _lib.ptr(t), n, k, h, w, mode, _lib.ptr(xy), _lib.ptr(mx), _lib.ptr(idx),
            _lib.current_stream(t.device)), 'decode')
    return xy, mx, idx


def soft_arg_max(batch_heatmaps):
    """Soft-arg-max over each [H,W] map: returns (preds [N,K,2] (x,y) in map
    pixels, maxvals [N,K,1] = raw maximum)."""
    xy, mx, _ = _decode(batch_heatmaps, 1, want_idx=False)
    return xy, mx


def soft_arg_max_np(batch_heatmaps):
    """img_proc.py:639-676: centre of mass with weights hm / sum(hm), zeroed where
    the maximum is <= 0.  numpy in -> numpy out like the reference (which, as a side
    effect not reproduced here, also normalises the caller's array in place)."""
    assert isinstance(batch_heatmaps, np.ndarray), 'batch_heatmaps should be numpy.ndarray'
    assert batch_heatmaps.ndim == 4, 'batch_images should be 4-ndim'
    t = torch.from_numpy(np.ascontiguousarray(batch_heatmaps, dtype=np.float32)).cuda()
    xy, mx, _ = _decode(t, 2, want_idx=False)
    return xy.cpu().numpy(), mx.cpu().numpy()


def hard_arg_max(batch_heatmaps):
    """(preds, maxvals, flat arg-max index [N,K] int32) on the device."""
    return _decode(batch_heatmaps, 0, want_idx=True)


def get_max_preds(batch_heatmaps):
    """Hard arg-max.  numpy in -> numpy out (the reference's contract); a CUDA
    tensor in -> CUDA tensors out."""
    if isinstance(batch_heatmaps, np.ndarray):
        assert batch_heatmaps.ndim == 4, 'batch_images should be 4-ndim'
        t = torch.from_numpy(np.ascontiguousarray(batch_heatmaps, dtype=np.float32)).cuda()
        xy, mx, _ = _decode(t, 0, want_idx=False)
        return xy.cpu().numpy(), mx.cpu().numpy()
    xy, mx, _ = _decode(batch_heatmaps, 0, want_idx=False)
    return xy, mx


# -- crop affine (host, per instance) ----------------------------------------

def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=0):
    """2x3 float64 affine of the crop defined by (center, scale*200 px, rotation in degrees)
    onto an ``output_size`` = (h, w) window -- img_proc.py:26-64.  Like there, the map is
    fixed by three point pairs held in float32 (centre; the point half a source width
    'above' it, rotated; the right-angle completion of the two) and solved exactly
    (cv2.getAffineTransform); ``inv`` swaps the roles (window -> image)."""
    scale_px = np.asarray(scale, dtype=np.float64) * SIZE
    center = np.asarray(center, dtype=np.float64)
    shift = np.asarray(shift, dtype=np.float64)
    dst_h, dst_w = output_size
    rad = np.pi * rot / 180
    up = -0.5 * scale_px[0]                                   # (0, -src_w/2) rotated by rad
    src_dir = np.array([-up * np.sin(rad), up * np.cos(rad)])
    tri = np.zeros((2, 3, 2), dtype=np.float32)               # [src|dst][point][xy]
    tri[0, 0] = center + scale_px * shift
    tri[0, 1] = center + src_dir + scale_px * shift
    tri[1, 0] = [dst_w * 0.5, dst_h * 0.5]
    tri[1, 1] = np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([0, dst_w * -0.5], np.float32)
    for t in tri:                                             # third point: b + perp(a - b)
        d = t[0] - t[1]
        t[2] = t[1] + np.array([-d[1], d[0]], dtype=np.float32)
    a, b = (tri[1], tri[0]) if inv else (tri[0], tri[1])
    m = np.hstack([a.astype(np.float64), np.ones((3, 1))])
    return np.linalg.solve(m, b.astype(np.float64)).T


def affine_transform_modified(pts, t):
    """[n,2] points through a 2x3 affine (img_proc.py:71-78)."""
    pts = np.asarray(pts)
    return (t @ np.hstack([pts, np.ones((len(pts), 1))]).T)[:2].T


# -- bounding-box helpers (host, per box) -----------------------------------
def enlarge_bbox(left, top, right, bottom, enlarge):
    cx, cy = (left + right) / 2, (top + bottom) / 2
    hw, hh = 0.5 * (right - left) * enlarge[0], 0.5 * (bottom - top) * enlarge[1]
    return [cx - hw, cy - hh, cx + hw, cy + hh]


def resize_bbox(left, top, right, bottom, target_ar=1.):
    width, height = right - left, bottom - top
    cx, cy = (left + right) / 2, (top + bottom) / 2
    if height / width > target_ar:
        half = 0.5 * height * (1 / target_ar)
        left, right = cx - half, cx + half
    else:
        half = 0.5 * width * target_ar
        top, bottom = cy - half, cy + half
    return {'bbox': [left, top, right, bottom], 'c': np.array([cx, cy]),
            's': np.array([(right - left) / SIZE, (bottom - top) / SIZE])}


def modify_bbox(bbox, target_ar, enlarge=1.1):
    box = enlarge_bbox(bbox[0], bbox[1], bbox[2], bbox[3], [enlarge, enlarge])
    return resize_bbox(box[0], box[1], box[2], box[3], target_ar=target_ar)


def generate_target_batch(joints, joints_vis, parameters, device=None, stream=None):
    """Heat-map targets of a whole batch on the GPU (csrc/targets.hip): the
    reference's per-sample ``generate_target`` (img_proc.py:347-409) for
    ``joints`` [N,K,>=2] (input-image pixels) and ``joints_vis`` [N,K].
    Returns CUDA tensors ``(target [N,K,hs[0],hs[1]], target_weight [N,K,1])``."""
    if parameters.get('target_type', 'gaussian') != 'gaussian':
        raise AssertionError('Only support gaussian map now!')
    if parameters.get('use_different_joints_weight'):
        raise NotImplementedError('use_different_joints_weight')
    L = _lib.lib()
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
    j = torch.as_tensor(np.asarray(joints) if not torch.is_tensor(joints) else joints)
    n, k = j.shape[:2]
    j3 = torch.zeros(n, k, 3, dtype=torch.float64, device=device)
    j3[..., :2] = j[..., :2].to(device=device, dtype=torch.float64)
    vis = torch.as_tensor(np.asarray(joints_vis) if not torch.is_tensor(joints_vis) else joints_vis)
    vis = vis.reshape(n, k).to(device=device, dtype=torch.float32).contiguous()
    inp, hs = parameters['input_size'], parameters['heatmap_size']
    rows, cols = int(hs[0]), int(hs[1])
    target = torch.empty(n, k, rows, cols, dtype=torch.float32, device=device)
    weight = torch.empty(n, k, 1, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        st = _lib.current_stream(device) if stream is None else stream
        _lib.check(L.egn_gaussian_targets_f32(_lib.ptr(j3), _lib.ptr(vis), n, k, rows, cols,
                                              float(inp[0]) / float(hs[0]), float(inp[1]) / float(hs[1]),
                                              float(parameters['sigma']), _lib.ptr(target), _lib.ptr(weight), st),
                   'gaussian targets')
    return target, weight


def generate_target(joints, joints_vis, parameters):
    """One sample, numpy in / numpy out like the reference (img_proc.py:347-409);
    computed on the GPU."""
    t, w = generate_target_batch(np.asarray(joints)[None], np.asarray(joints_vis)[None], parameters)
    return t[0].cpu().numpy(), w[0].cpu().numpy()


def to_npy(tensor):
    return tensor if isinstance(tensor, np.ndarray) else tensor.data.cpu().numpy()
