"""get_distance_src with the heat-map decode on the GPU, and the numpy-style
soft-arg-max (decode mode 2), against outputs of the REFERENCE (metric.npz, decode.npz)."""
import numpy as np
import pytest
import torch

from conftest import golden
from egonet_amd.common import img_proc
from egonet_amd.metric import criterions

pytestmark = pytest.mark.gpu


def _meta(g, rotation=True):
    m = {'center': g['center'], 'scale': g['scale'], 'original_joints': g['original_joints']}
    if rotation:
        m['rotation'] = g['rotation']
    return m


@pytest.mark.parametrize('tag,as_tensor', [('hard', False), ('hard', True), ('soft', False)])
def test_get_distance_src_heatmaps_vs_reference(tag, as_tensor):
    g = golden('metric.npz')
    out = torch.from_numpy(g['heatmaps']).cuda() if as_tensor else g['heatmaps'].copy()
    avg, cnt, others = criterions.get_distance_src(out, _meta(g), arg_max=tag, image_size=(64.0, 64.0))
    assert cnt == int(g[tag + '/cnt'])
    # key-points within 1e-3 px of the reference (heat-map pixels x stride 4 x crop scale)
    np.testing.assert_allclose(others['joints_pred'], g[tag + '/joints_pred'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(others['src_coord'], g[tag + '/src_coord'], rtol=0, atol=5e-3)
    np.testing.assert_allclose(avg, float(g[tag + '/avg']), rtol=1e-5)
    np.testing.assert_array_equal(others['correct_cnt'], g[tag + '/correct_cnt'])
    assert (others['joints_pred'][0, 0] == 0).all()                 # all-negative map: zeroed


def test_get_distance_src_without_rotation_key():
    g = golden('metric.npz')
    avg, _, others = criterions.get_distance_src(g['heatmaps'].copy(), _meta(g, rotation=False), arg_max='hard',
                                                 image_size=(64.0, 64.0))
    np.testing.assert_allclose(others['src_coord'], g['norot/src_coord'], rtol=0, atol=5e-3)
    np.testing.assert_allclose(avg, float(g['norot/avg']), rtol=1e-5)


def test_tensor_without_arg_max_mode_is_refused():
    with pytest.raises(NotImplementedError):
        criterions.get_distance_src(torch.zeros(1, 2, 4, 4).cuda(), {'center': [], 'scale': []}, arg_max=None)


def test_soft_arg_max_np_kernel_vs_oracle_and_reference():
    from oracle import decode_oracle
    g = golden('metric.npz')
    hm = g['heatmaps']
    got, mv = img_proc.soft_arg_max_np(hm.copy())
    want, wmv = decode_oracle.soft_arg_max_np(hm.copy())
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-3)
    np.testing.assert_array_equal(mv, wmv)
    np.testing.assert_allclose(got * 4.0, g['soft/joints_pred'], rtol=0, atol=1e-3)
