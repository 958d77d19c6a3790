"""Crop front end (csrc/crop.hip, egonet_amd/common/crop_gpu.py, oracle/crop_oracle.py).

cv2 is absent from the image, so the oracle (a restatement of cv::warpAffine's
published 8-bit bilinear scheme) is UNPINNED against cv2 itself; the CPU tests
here check it against what any correct bilinear warp must satisfy and against
scipy's float bilinear sampler, the GPU tests check the kernel against the
oracle bit for bit."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import crop_oracle, geometry_oracle


def _image(h=96, w=160, seed=0, smooth=False):
    rng = np.random.RandomState(seed)
    if smooth:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 40 + xx * 1.1, 200 - yy * 1.5], axis=2)
        return np.clip(img, 0, 255).astype(np.uint8)
    return rng.randint(0, 256, (h, w, 3)).astype(np.uint8)


def test_identity_and_integer_translation_are_exact():
    img = _image()
    h, w = img.shape[:2]
    eye = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(crop_oracle.warp_affine_u8(img, eye, (w, h)), img)
    shift = np.array([[1.0, 0, -7.0], [0, 1.0, 5.0]])            # crop(x, y) = img(x + 7, y - 5)
    out = crop_oracle.warp_affine_u8(img, shift, (w, h))
    assert np.array_equal(out[5:, :w - 7], img[:h - 5, 7:])
    assert not out[:5].any() and not out[:, w - 7:].any()         # BORDER_CONSTANT 0


def test_half_pixel_shift_averages_neighbours():
    img = _image(seed=1)
    out = crop_oracle.warp_affine_u8(img, np.array([[1.0, 0, -0.5], [0, 1.0, 0]]), (img.shape[1], img.shape[0]))
    want = (img[:, :-1].astype(np.int64) + img[:, 1:].astype(np.int64) + 1) >> 1       # (a + b) / 2 rounded half up
    assert np.array_equal(out[:, :-1], want.astype(np.uint8))


def test_against_scipy_float_bilinear_on_a_smooth_image():
    img = _image(smooth=True)
    box = np.array([31.0, 20.0, 120.0, 77.0])
    ret = geometry_oracle.modify_bbox(box, 1.0)
    M = crop_oracle.forward_affine(ret['c'], ret['s'], (64, 64))
    got = crop_oracle.warp_affine_u8(img, M, (64, 64)).astype(np.float64)
    k, tx, ty = M[0, 0], M[0, 2], M[1, 2]
    ys, xs = np.mgrid[0:64, 0:64]
    src = np.stack([(ys - ty) / k, (xs - tx) / k])
    want = np.stack([ndimage.map_coordinates(img[..., c].astype(np.float64), src, order=1, mode='constant', cval=0.0)
                     for c in range(3)], axis=2)
    inside = (src[0] > 1) & (src[0] < img.shape[0] - 2) & (src[1] > 1) & (src[1] < img.shape[1] - 2)
    # 1/32-pixel coordinate quantisation x the image's gradient (< 12 levels / pixel) + rounding
    assert np.abs(got - want)[inside].max() < 1.0


def test_forward_affine_is_the_reference_crop_transform():
    """The closed form against the oracle's restatement of get_affine_transform (inv=1),
    itself pinned to the reference in tests/golden/egonet_pipeline.npz."""
    c, s = np.array([211.3, 140.2]), np.array([0.93, 0.93])
    fwd = np.vstack([crop_oracle.forward_affine(c, s, (256, 256)), [0, 0, 1]])
    inv = np.vstack([geometry_oracle.inverse_crop_affine(c, s, (256, 256)), [0, 0, 1]])
    np.testing.assert_allclose(fwd @ inv, np.eye(3), atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('seed,hw', [(0, (96, 160)), (3, (375, 1242))])
def test_gpu_crops_equal_the_oracle(seed, hw):
    from egonet_amd.common import crop_gpu
    from egonet_amd import synth
    img = _image(hw[0], hw[1], seed=seed)
    boxes = synth.synth_boxes(7, seed=seed)
    boxes[0] = [-20.0, -10.0, 60.0, 50.0]                     # sticks out of the image: zero border
    boxes[1] = [hw[1] - 40.0, hw[0] - 30.0, hw[1] + 25.0, hw[0] + 9.0]
    rets = [geometry_oracle.modify_bbox(b, 1.0) for b in boxes]
    centers, scales = [r['c'] for r in rets], [r['s'] for r in rets]
    mean, std = crop_gpu.IMAGENET_MEAN, crop_gpu.IMAGENET_STD
    want = crop_oracle.crop_instances(img, centers, scales, (128, 128), mean, std)
    got = crop_gpu.crop_boxes(img, centers, scales, (128, 128), mean, std)
    assert tuple(got.shape) == (7, 3, 128, 128)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-6)
    # the uint8 levels behind the floats are identical
    lvl = np.rint((got.cpu().numpy() * np.float32(std).reshape(3, 1, 1) + np.float32(mean).reshape(3, 1, 1)) * 255)
    lvl_w = np.rint((want * np.float32(std).reshape(3, 1, 1) + np.float32(mean).reshape(3, 1, 1)) * 255)
    assert np.array_equal(lvl, lvl_w)


@pytest.mark.gpu
def test_egonet_forward_from_an_image_array():
    """annot_dict + uint8 frame -> records with 2D / 3D key-points; the crops feed the
    same batched pipeline infer_crops() runs (checked against it)."""
    from egonet_amd import configs, synth
    from egonet_amd.common import crop_gpu
    from egonet_amd.model.egonet import EgoNet
    cfg = configs.hrnet_config(8, (64, 64), 33, 'coordinates', modules=(1, 1, 1), num_blocks=1, lifter_neurons=128)
    ego = EgoNet(cfg, pre_trained=False)
    ego.HC.load_state_dict(synth.synth_state_dict(ego.HC.state_dict(), seed=6))
    ego.L.load_state_dict(synth.synth_state_dict(ego.L.state_dict(), seed=7))
    ego.LS = synth.synth_lifter_stats(66, 96, seed=1)
    ego = ego.eval().cuda()
    img = _image(375, 1242, seed=5, smooth=True)
    boxes = synth.synth_boxes(4, seed=2)
    annot = {'path': ['mem://frame0.png'], 'boxes': [boxes]}
    recs = ego.forward(annot, images={'mem://frame0.png': img})
    r = recs['mem://frame0.png']
    assert len(r['kpts_2d_pred']) == 4 and r['kpts_3d_pred'].shape == (4, 32, 3)
    rets = [geometry_oracle.modify_bbox(b, 1.0) for b in boxes]
    crops = crop_gpu.crop_boxes(img, [t['c'] for t in rets], [t['s'] for t in rets], (64, 64))
    res = ego.infer_crops(crops, np.stack([t['c'] for t in rets]), np.stack([t['s'] for t in rets]))
    # two routes through the same kernels (per-image records vs one batched program)
    np.testing.assert_allclose(np.concatenate(r['kpts_2d_pred']), res['kpts_2d'], atol=1e-3)
    np.testing.assert_allclose(r['kpts_3d_pred'], res['kpts_3d'], atol=1e-3)
