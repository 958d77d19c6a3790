"""The exact float64 code of the geometry kernels (csrc/pose_math.h), compiled
for the host with g++, against the reference-generated fixtures."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import golden, ROOT

BUILD = os.path.join(ROOT, 'tests', '_build')


@pytest.fixture(scope='module')
def harness():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, 'host_math_harness.so')
    src = os.path.join(ROOT, 'tests', 'host_math_harness.cpp')
    subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-o', so, src])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _pose(h, preds, kx, K, mode):
    n = len(preds)
    p = np.ascontiguousarray(preds.reshape(n, -1), dtype=np.float64)
    e, a = np.zeros((n, 3)), np.zeros(n)
    kx = np.ascontiguousarray(kx, dtype=np.float64)
    h.harness_pose(_p(p), n, _p(kx), C.c_double(K[0, 0]), C.c_double(K[0, 2]), mode, _p(e), _p(a))
    return e, a


def test_pose_solve_cuboids(harness):
    g = golden('pose_solve.npz')
    e, a = _pose(harness, g['preds'], g['kpts_x'], g['K'], 0)
    np.testing.assert_allclose(e, g['euler'], atol=1e-9)
    np.testing.assert_allclose(a, g['alpha_proj'], atol=1e-9)
    e, a = _pose(harness, g['preds'], g['kpts_x'], g['K'], 1)
    np.testing.assert_allclose(a, g['alpha_trans'], atol=1e-9)


def test_pose_solve_on_network_outputs(harness):
    """Ill-conditioned case: lifted cuboids of a random-weight lifter."""
    g = golden('egonet_pipeline.npz')
    e, a = _pose(harness, g['kpts_3d'], g['kpts_2d'][:, 0], g['K'], 0)
    np.testing.assert_allclose(np.cos(e), np.cos(g['euler']), atol=1e-8)
    np.testing.assert_allclose(np.sin(e), np.sin(g['euler']), atol=1e-8)
    np.testing.assert_allclose(np.cos(a), np.cos(g['alpha_proj']), atol=1e-8)
    np.testing.assert_allclose(np.sin(a), np.sin(g['alpha_proj']), atol=1e-8)


def test_crop_affine_matches_reference_solve(harness):
    """Closed form vs the 6x6 solve on the reference's float32 control points
    (oracle.geometry_oracle.inverse_crop_affine) for many boxes."""
    from oracle import geometry_oracle as go
    from egonet_amd import synth
    boxes = synth.synth_boxes(200, seed=9)
    rng = np.random.RandomState(3)
    for crop_w, crop_h in ((256, 256), (192, 256)):
        local = rng.uniform(0, 1, (200, 33, 2)).astype(np.float32)
        rets = [go.modify_bbox(b, crop_h / crop_w) for b in boxes]
        c = np.stack([r['c'] for r in rets])
        s = np.stack([r['s'] for r in rets])
        out = np.zeros((200, 33, 2))
        harness.harness_crop_to_screen(_p(local), 200, 33, C.c_double(crop_w), C.c_double(crop_h), _p(c), _p(s),
                                       crop_w, crop_h, _p(out))
        for i in range(200):
            loc = (local[i] * np.array([crop_w, crop_h]).reshape(1, 2)).astype(np.float32)
            want = go.crop_to_screen(loc, c[i], s[i], (crop_h, crop_w))
            np.testing.assert_allclose(out[i], want, rtol=0, atol=1e-9)
