"""Instance cropping with OpenCV -- the front end the reference runs before the
hot path (libs/model/egonet.py:68-155).  Not part of the accelerated path
(SURVEY.md section 8f, rank 1); kept so that ``EgoNet.forward(annot_dict)``
works when ``cv2`` is installed.  Needs ``model.pth_trans`` (the torchvision
ToTensor+Normalize pipeline the caller sets, tools/inference.py:147).
"""
import numpy as np
import torch


def _affine_fwd(center, scale, out_wh):
    """2x3 screen->crop affine for rot=0 (img_proc.py:26-64, inv=0)."""
    w, h = out_wh
    src_w = scale[0] * 200.0
    k = w / src_w
    return np.array([[k, 0.0, w * 0.5 - k * center[0]],
                     [0.0, k, h * 0.5 - k * center[1]]], dtype=np.float64)


def crop_instances(model, annot_dict):
    import cv2
    width, height = model.resolution
    records = model.make_records(annot_dict)
    cache, crops = {}, []
    for rec in records:
        path = rec['path']
        if path not in cache:
            img = cv2.imread(path, 1 | 128)
            if img is None:
                raise ValueError('Fail to read {}'.format(path))
            cache[path] = cv2.cvtColor(img, cv2.COLOR_BGR2RGB)
        trans = _affine_fwd(rec['center'], rec['scale'], (width, height))
        patch = cv2.warpAffine(cache[path], trans, (int(width), int(height)), flags=cv2.INTER_LINEAR)
        patch = patch if model.pth_trans is None else model.pth_trans(patch)
        if not torch.is_tensor(patch):
            patch = torch.from_numpy(np.ascontiguousarray(patch)).permute(2, 0, 1).float()
        crops.append(patch.unsqueeze(0))
    return torch.cat(crops, dim=0), records
