"""CPU oracle: crop geometry, lifter normalisation and pose solve
(TEST INFRASTRUCTURE ONLY).

Reference followed (paths relative to /root/reference):
  * modify_bbox / enlarge_bbox / resize_bbox   libs/common/img_proc.py:411-459
  * get_affine_transform(inv=1)                libs/common/img_proc.py:26-64
    (``cv2.getAffineTransform`` is third-party OpenCV 3.4.2, not vendored in
    the reference; for 3 exact point pairs it returns the unique affine map,
    restated here as a 6x6 float64 linear solve)
  * affine_transform_modified                  libs/common/img_proc.py:71-78
  * normalize_1d / unnormalize_1d              libs/dataset/normalization/operations.py:20-51
  * get_template / kpts_to_euler / get_6d_rep  libs/model/egonet.py:238-295
  * compute_rigid_transform (Kabsch)           libs/common/transformation.py:99-134
  * get_observation_angle_proj / _trans        libs/model/egonet.py:203-236
  * interp_dict['bbox12']                      libs/dataset/KITTI/car_instance.py:63-70
"""
import math
import numpy as np
from scipy.spatial.transform import Rotation

SIZE = 200.0
# cuboid edges (1-based corner ids): 4 along h, 4 along l, 4 along w
EDGE_PARENT = np.array([1, 3, 5, 7, 1, 2, 3, 4, 1, 2, 5, 6])
EDGE_CHILD = np.array([2, 4, 6, 8, 5, 6, 7, 8, 3, 4, 7, 8])


def modify_bbox(bbox, target_ar, enlarge=1.1):
    """img_proc.py:411-459.  Returns dict(bbox, c, s) like the reference."""
    l, t, r, b = [float(v) for v in bbox[:4]]
    w, h = r - l, b - t
    cx, cy = (l + r) / 2, (t + b) / 2
    w, h = w * enlarge, h * enlarge
    l, r, t, b = cx - 0.5 * w, cx + 0.5 * w, cy - 0.5 * h, cy + 0.5 * h
    w, h = r - l, b - t
    cx, cy = (l + r) / 2, (t + b) / 2
    if h / w > target_ar:
        nw = h * (1 / target_ar)
        l, r = cx - 0.5 * nw, cx + 0.5 * nw
    else:
        nh = w * target_ar
        t, b = cy - 0.5 * nh, cy + 0.5 * nh
    return {'bbox': [l, t, r, b], 'c': np.array([cx, cy]),
            's': np.array([(r - l) / SIZE, (b - t) / SIZE])}


def _solve_affine(src, dst):
    """Unique 2x3 affine A with A @ [src;1] = dst for 3 point pairs (float64)."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    m = np.hstack([src, np.ones((3, 1))])
    return np.linalg.solve(m, dst).T           # [2,3]


def inverse_crop_affine(center, scale, out_hw):
    """get_affine_transform(center, scale, rot=0, (H,W), inv=1):
    crop pixels -> screen pixels (img_proc.py:26-64, float32 control points)."""
    center = np.asarray(center, dtype=np.float64)
    scale_tmp = np.asarray(scale, dtype=np.float64) * SIZE
    src_w = scale_tmp[0]
    dst_h, dst_w = out_hw
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0] = center
    src[1] = center + np.array([0.0, src_w * -0.5])
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([0, dst_w * -0.5], np.float32)
    for pts in (src, dst):
        d = pts[0] - pts[1]
        pts[2] = pts[1] + np.array([-d[1], d[0]], dtype=np.float32)
    return _solve_affine(dst, src)


def crop_to_screen(local_xy, center, scale, out_hw):
    """egonet.py:443-453: apply the inverse crop affine to [K,2] points."""
    t = inverse_crop_affine(center, scale, out_hw)
    pts = np.hstack([local_xy, np.ones((len(local_xy), 1))]).T
    return (t @ pts)[:2].T


def normalize_1d(data, mean, std):
    return (data - mean) / std


def unnormalize_1d(data, mean, std):
    return data * std + mean


def cuboid_template(pred, interp_coef=(0.332, 0.667)):
    """egonet.py:238-263.  pred [32,3] (root dropped) -> template [3,32]."""
    edges = pred[EDGE_PARENT - 1] - pred[EDGE_CHILD - 1]
    length = np.sqrt((edges ** 2).sum(axis=1))
    h, l, w = length[:4].sum() / 4, length[4:8].sum() / 4, length[8:].sum() / 4
    xc = np.array([l, l, l, l, 0, 0, 0, 0], dtype=np.float64) - np.float32(l) / 2
    yc = np.array([0, h, 0, h, 0, h, 0, h], dtype=np.float64) - np.float32(h)
    zc = np.array([w, w, 0, 0, w, w, 0, 0], dtype=np.float64) - np.float32(w) / 2
    corners = np.array([xc, yc, zc])
    if len(pred) == 32:
        par, chi = corners[:, EDGE_PARENT - 1], corners[:, EDGE_CHILD - 1]
        corners = np.hstack([corners] + [par + c * (chi - par) for c in interp_coef])
    return corners


def rigid_transform(x, y):
    """transformation.py:99-134 (unweighted): R,t minimising |R x + t - y|."""
    cx, cy = x.mean(axis=1, keepdims=True), y.mean(axis=1, keepdims=True)
    hmat = (x - cx) @ (y - cy).T
    u, _, vt = np.linalg.svd(hmat)
    r = vt.T @ u.T
    if np.linalg.det(r) < 0:
        vt[-1, :] *= -1
        r = vt.T @ u.T
    return r, -r @ cx + cy


def six_dof(pred3d):
    """egonet.py:279-295: [n,32,3] -> euler [n,3] (x,y,z order), translation [n,3]."""
    pred3d = pred3d.reshape(len(pred3d), -1, 3)
    angles = []
    for p in pred3d:
        r, _ = rigid_transform(cuboid_template(p), p.T)
        a = Rotation.from_matrix(r).as_euler('yxz', degrees=False)
        angles.append(a[[1, 0, 2]].reshape(1, 3))
    return np.concatenate(angles), pred3d[:, 0, :]


def _wrap(alpha):
    while alpha > math.pi:
        alpha -= 2 * math.pi
    while alpha < -math.pi:
        alpha += 2 * math.pi
    return alpha


def observation_angle_proj(euler, kpts_x0, K):
    """egonet.py:219-236: alpha from the projected centre x (first key-point)."""
    f, cx = K[0, 0], K[0, 2]
    return np.array([_wrap(euler[i][1] - math.atan2(-f, kpts_x0[i] - cx) - 0.5 * math.pi)
                     for i in range(len(euler))])


def observation_angle_trans(euler, trans):
    """egonet.py:203-217."""
    return np.array([_wrap(euler[i][1] - math.atan2(-trans[i][2], trans[i][0]) - 0.5 * math.pi)
                     for i in range(len(euler))])
