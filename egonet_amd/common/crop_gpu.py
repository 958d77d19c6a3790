"""GPU crop front end (csrc/crop.hip): the reference's ``crop_instances`` /
``crop_single_instance`` (libs/model/egonet.py:68-155) with the warp, ToTensor and
Normalize of ALL boxes of an image in one launch.

The uint8 image is uploaded once (1.4 MB for a KITTI frame instead of 786 KB of
fp32 per crop); the per-box host work that remains is ``modify_bbox`` (a dozen
flops).  Images are read with PIL (RGB order, what the reference gets after its
BGR->RGB swap, egonet.py:98-104); arrays can be passed directly.
"""
import numpy as np
import torch

from .. import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # configs/KITTI_inference:demo.yml:52-53
IMAGENET_STD = (0.229, 0.224, 0.225)


def forward_affines(centers, scales, out_wh):
    """[n,6] float64 image -> crop affines for rot = 0 (get_affine_transform, img_proc.py:26-64)."""
    w, h = out_wh
    c = np.asarray(centers, dtype=np.float64).reshape(-1, 2)
    s = np.asarray(scales, dtype=np.float64).reshape(-1, 2)
    k = w / (s[:, 0] * 200.0)
    m = np.zeros((len(c), 6), dtype=np.float64)
    m[:, 0], m[:, 2] = k, w * 0.5 - k * c[:, 0]
    m[:, 4], m[:, 5] = k, h * 0.5 - k * c[:, 1]
    return m


_CONST = {}


def _norm_consts(mean, std, device):
    """mean / std as device tensors, uploaded once per (device, values): a per-call ``torch.tensor(..., device=)`` is a
    blocking copy from pageable memory -- two per frame serialised a serving loop that overlaps uploads with compute."""
    key = (str(device), tuple(float(v) for v in mean), tuple(float(v) for v in std))
    c = _CONST.get(key)
    if c is None:
        c = (torch.tensor(key[1], dtype=torch.float32, device=device), torch.tensor(key[2], dtype=torch.float32, device=device))
        _CONST[key] = c
    return c


def crop_boxes(img, centers, scales, out_wh, mean=IMAGENET_MEAN, std=IMAGENET_STD, device=None, affines=None, out=None):
    """img [H,W,3] uint8 RGB (numpy or tensor, host or device) -> CUDA [n,3,h,w] fp32 crops.
    ``affines``: the boxes' [n,6] float64 image -> crop affines already on the device (``forward_affines`` uploaded by
    the caller, e.g. once per batch from pinned memory) -- then ``centers`` / ``scales`` are not read and the call issues
    no host -> device copy of its own; ``out``: a preallocated [n,3,h,w] fp32 device tensor to write into."""
    L = _lib.lib()
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
    t = torch.as_tensor(img)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError('crop_boxes expects an [H,W,3] uint8 RGB image, got %s %s' % (t.dtype, tuple(t.shape)))
    t = t.to(device).contiguous()
    H, W = t.shape[:2]
    w, h = int(out_wh[0]), int(out_wh[1])
    if affines is None:
        M = torch.from_numpy(forward_affines(centers, scales, (w, h))).to(device)
    else:
        M = affines
        if not (M.is_cuda and M.dtype == torch.float64 and M.dim() == 2 and M.shape[1] == 6 and M.is_contiguous()):
            raise ValueError('affines: a contiguous [n,6] float64 CUDA tensor')
    n = M.shape[0]
    if out is None:
        out = torch.empty(n, 3, h, w, dtype=torch.float32, device=device)
    elif tuple(out.shape) != (n, 3, h, w) or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous():
        raise ValueError('out: a contiguous [n,3,h,w] fp32 CUDA tensor')
    if n == 0:
        return out
    mean_t, std_t = _norm_consts(mean, std, device)
    with torch.cuda.device(device):
        _lib.check(L.egn_crop_warp_normalize_u8(_lib.ptr(t), H, W, 3 * W, _lib.ptr(M), n, h, w, _lib.ptr(mean_t),
                                                _lib.ptr(std_t), _lib.ptr(out), _lib.current_stream(device)), 'crop')
    return out


def load_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert('RGB'))          # a writable copy (torch.as_tensor warns otherwise)


def crop_instances(model, annot_dict, images=None):
    """-> (instances [n,3,h,w] CUDA, records) like egonet.py:105-155.  ``images``:
    optional {path: [H,W,3] uint8 RGB}; otherwise the files are read with PIL."""
    width, height = model.resolution
    if model.xy_dict is not None and model.xy_dict.get('flag'):
        # the reference concatenates the x/y ramps to the crop (egonet.py:88-93); the device
        # front end produces the 3 image channels only
        raise NotImplementedError('add_xy: the GPU crop front end writes 3-channel crops; set pth_trans to use '
                                  'the host route (common/crop_cv2.py) for 5-channel inputs')
    records = model.make_records(annot_dict)
    dev = next(model.parameters()).device
    norm = (model.cfgs.get('dataset', {}) or {}).get('pth_transform') or {}
    mean, std = list(norm.get('mean', IMAGENET_MEAN)), list(norm.get('std', IMAGENET_STD))
    if len(mean) != 3 or len(std) != 3:
        raise ValueError('pth_transform mean/std must have 3 entries (RGB), got %d / %d' % (len(mean), len(std)))
    by_path = {}
    for i, rec in enumerate(records):
        by_path.setdefault(rec['path'], []).append(i)
    crops = [None] * len(records)
    for path, idxs in by_path.items():
        img = images[path] if images is not None and path in images else load_rgb(path)
        out = crop_boxes(img, [records[i]['center'] for i in idxs], [records[i]['scale'] for i in idxs],
                         (width, height), mean, std, dev)
        for j, i in enumerate(idxs):
            crops[i] = out[j:j + 1]
    if not crops:
        return torch.empty(0, 3, int(height), int(width), device=dev), records
    return torch.cat(crops, dim=0), records
