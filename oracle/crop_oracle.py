"""CPU oracle: instance crop = affine warp (bilinear) + ToTensor + Normalize
(TEST INFRASTRUCTURE ONLY).  **PARITY UNPINNED.**

Reference call site: libs/model/egonet.py:68-96 (crop_single_instance):
``cv2.warpAffine(img, get_affine_transform(c, s, 0, (h, w)), (w, h), flags=INTER_LINEAR)``
followed by torchvision ``ToTensor`` + ``Normalize`` (car_instance.py:522-531).

cv2 is a third-party dependency (OpenCV 3.4.2, pinned in docs/spec-list.txt) that
is neither vendored in the reference nor installed in the build image, and the
reference holds no test vector for it, so this restatement of the published
algorithm of ``cv::warpAffine`` for 8-bit ``INTER_LINEAR`` (OpenCV
modules/imgproc/src/imgwarp.cpp) cannot be checked against cv2 here:
  * the matrix is inverted in double (no WARP_INVERSE_MAP);
  * source coordinates in fixed point, AB_BITS = 10: the per-column term
    round(M0*x*1024) and the per-row term round((M1*y+M2)*1024) + 16 are rounded
    separately (half to even), summed, and shifted to 1/32 pixel (INTER_BITS 5);
  * bilinear weights (32-fx)(32-fy)/1024 etc. as 15-bit integers -- exact, so
    the weight table's sum fix-up never triggers; value = (sum + 2^14) >> 15;
  * BORDER_CONSTANT with value 0, applied per tap.
The affine itself (rot = 0) is the closed form of get_affine_transform, pinned
to <= 8e-5 px against the reference (tests/golden/egonet_pipeline.npz).
"""
import numpy as np


def forward_affine(center, scale, out_wh):
    """Image -> crop 2x3 affine for rot = 0 (img_proc.py:26-64)."""
    w, h = out_wh
    k = w / (scale[0] * 200.0)
    return np.array([[k, 0.0, w * 0.5 - k * center[0]], [0.0, k, h * 0.5 - k * center[1]]], dtype=np.float64)


def warp_affine_u8(img, M, out_wh):
    """img [H,W,C] uint8 -> [out_h,out_w,C] uint8."""
    ow, oh = out_wh
    H, W = img.shape[:2]
    m = np.asarray(M, dtype=np.float64).reshape(2, 3)
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    i0, i1, i3, i4 = m[1, 1] * D, -m[0, 1] * D, -m[1, 0] * D, m[0, 0] * D
    i2 = -i0 * m[0, 2] - i1 * m[1, 2]
    i5 = -i3 * m[0, 2] - i4 * m[1, 2]
    xs, ys = np.arange(ow, dtype=np.float64), np.arange(oh, dtype=np.float64)
    X = (np.rint((i1 * ys + i2) * 1024.0).astype(np.int64)[:, None] + 16 + np.rint(i0 * xs * 1024.0).astype(np.int64)[None]) >> 5
    Y = (np.rint((i4 * ys + i5) * 1024.0).astype(np.int64)[:, None] + 16 + np.rint(i3 * xs * 1024.0).astype(np.int64)[None]) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return v * ok[..., None]
    acc = (((32 - fx) * (32 - fy))[..., None] * tap(sy, sx) + (fx * (32 - fy))[..., None] * tap(sy, sx + 1)
           + ((32 - fx) * fy)[..., None] * tap(sy + 1, sx) + (fx * fy)[..., None] * tap(sy + 1, sx + 1))
    return ((acc + 512) >> 10).astype(np.uint8)


def crop_instances(img, centers, scales, out_wh, mean, std):
    """-> [n,3,h,w] float32 normalised crops (what the backbone consumes)."""
    mean = np.asarray(mean, dtype=np.float32).reshape(3, 1, 1)
    std = np.asarray(std, dtype=np.float32).reshape(3, 1, 1)
    outs = []
    for c, s in zip(centers, scales):
        patch = warp_affine_u8(img, forward_affine(c, s, out_wh), out_wh)
        t = patch.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        outs.append((t - mean) / std)
    return np.stack(outs).astype(np.float32)
