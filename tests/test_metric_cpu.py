"""Host side of the validation metric (egonet_amd.metric.criterions) against outputs of
the REFERENCE's get_distance_src (tests/golden/metric.npz, made by make_golden.py
section 0e).  The coordinate-head path needs no decode, so it runs without a GPU."""
import numpy as np
import torch

from conftest import golden
from egonet_amd.common import img_proc
from egonet_amd.metric import criterions


def _meta(g, rotation=True):
    m = {'center': g['center'], 'scale': g['scale'], 'original_joints': g['original_joints']}
    if rotation:
        m['rotation'] = g['rotation']
    return m


def test_get_distance_src_coordinate_head_vs_reference():
    g = golden('metric.npz')
    out = (torch.from_numpy(g['heatmaps']), torch.from_numpy(g['coords'].copy()))
    avg, cnt, others = criterions.get_distance_src(out, _meta(g), arg_max='hard', image_size=(64.0, 64.0))
    assert cnt == int(g['coords/cnt']) == int((g['original_joints'][..., 2] != 0).sum())
    np.testing.assert_allclose(others['src_coord'], g['coords/src_coord'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(avg, float(g['coords/avg']), rtol=1e-9)
    np.testing.assert_array_equal(others['correct_cnt'], g['coords/correct_cnt'])
    np.testing.assert_allclose(others['joints_pred'], g['coords/joints_pred'], rtol=0, atol=1e-6)
    assert others['joints_pred'].shape[0] == 5 and others['src_coord'].shape[0] == 4      # one unlabeled extra
    np.testing.assert_allclose(others['PCK_batch'], g['coords/correct_cnt'] / cnt)


def test_affine_is_a_similarity_and_inverts():
    rng = np.random.RandomState(0)
    for rot in (0.0, 17.0, -45.0, 90.0):
        c, s = rng.uniform(100, 900, 2), np.repeat(rng.uniform(0.3, 2.0), 2)
        fwd = img_proc.get_affine_transform(c, s, rot, (256, 256))
        inv = img_proc.get_affine_transform(c, s, rot, (256, 256), inv=1)
        a = np.vstack([fwd, [0, 0, 1]]) @ np.vstack([inv, [0, 0, 1]])
        np.testing.assert_allclose(a, np.eye(3), atol=2e-4)
        lin = fwd[:, :2]
        k = 256.0 / (s[0] * 200.0)
        np.testing.assert_allclose(lin @ lin.T, k * k * np.eye(2), rtol=1e-5, atol=1e-6)   # float32 points
        np.testing.assert_allclose(fwd @ np.array([c[0], c[1], 1.0]), [128.0, 128.0], atol=1e-3)
    # rot = 0: the closed form of EgoNet.get_keypoints, X = cx + (u - 128) * 200 s / 256
    c, s = np.array([400.0, 300.0]), np.array([0.8, 0.8])
    inv = img_proc.get_affine_transform(c, s, 0, (256, 256), inv=1)
    pts = np.array([[0.0, 0.0], [128.0, 128.0], [255.0, 10.0]])
    want = c + (pts - 128.0) * (200.0 * 0.8 / 256.0)
    np.testing.assert_allclose(img_proc.affine_transform_modified(pts, inv), want, atol=1e-4)


def test_distance_and_pck_helpers():
    gt = np.array([[0.0, 0.0, 1.0], [10.0, 30.0, 0.0], [4.0, 3.0, 1.0]])
    pred = np.array([[3.0, 4.0], [0.0, 0.0], [4.0, 3.0]])
    assert criterions.get_distance(gt, pred) == [5.0, 0.0]                 # invisible joint skipped
    assert len(criterions.get_distance(gt[:, :2], pred)) == 3
    # extent 30 px / 3 = 10: thresholds 1, 2, 3 px -> only the exact hit counts
    np.testing.assert_array_equal(criterions.get_PCK(pred, gt), [1, 1, 1])
    err, cnt, _ = criterions.get_angle_error(np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, -1e-9]]),
                                             {'angles_gt': np.array([0.0, 0.0, np.pi])})
    assert cnt == 3 and abs(err - (0 + 90 + 0) / 3) < 1e-5                 # wrap-around at 180 deg


def test_running_metric_accumulates():
    g = golden('metric.npz')
    cfgs = {'heatmapModel': {'num_joints': 33, 'input_size': [64.0, 64.0]}, 'testing_settings': {}}
    m = criterions.JointDistance2DSIP(cfgs)
    out = (torch.from_numpy(g['heatmaps']), torch.from_numpy(g['coords'].copy()))
    m.update(out, _meta(g))
    m.update(out, _meta(g))
    assert m.count == 2 * int(g['coords/cnt'])
    np.testing.assert_allclose(m.mean, float(g['coords/avg']), rtol=1e-9)
    np.testing.assert_array_equal(m.PCK_counts, 2 * g['coords/correct_cnt'])
