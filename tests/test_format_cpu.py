"""KITTI result writer (egonet_amd/common/format.py) against strings produced by the
reference's ``libs/common/format.py`` (tests/golden/format.json) -- byte for byte."""
import json
import os

import numpy as np
import pytest

from egonet_amd.common import format as fmt

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    with open(os.path.join(HERE, 'golden', 'format.json')) as f:
        return json.load(f)


def _record(c):
    return {'raw_txt_format': c['raw_txt_format'], 'euler_angles': np.array(c['euler_angles']),
            'alphas': np.array(c['alphas'])}


def test_pred_str_equals_reference_bytes():
    cases = _cases()
    assert [len(c['raw_txt_format']) for c in cases] == [1, 3, 4]
    for c in cases:
        assert fmt.get_pred_str(_record(c)) == c['pred_str']
        assert not c['pred_str'].endswith('\n') and c['pred_str'].count('\n') == len(c['alphas']) - 1


def test_prediction_does_not_modify_the_input_annotations():
    c = _cases()[1]
    rec = _record(c)
    before = json.dumps(rec['raw_txt_format'])
    fmt.get_pred_str(rec)
    assert json.dumps(rec['raw_txt_format']) == before            # reference deep-copies (format.py:49)


def test_label_line_round_trip():
    for c in _cases():
        for line, raw, ang, alpha in zip(c['pred_str'].split('\n'), c['raw_txt_format'], c['euler_angles'], c['alphas']):
            d = fmt.parse_label_line(line)
            assert d['class'] == raw['class']
            np.testing.assert_allclose(d['dimensions'], raw['dimensions'], atol=5e-7)      # (l, h, w) order restored
            np.testing.assert_allclose(d['locations'], raw['locations'], atol=5e-7)
            np.testing.assert_allclose(d['bbox'], raw['bbox'], atol=5e-7)
            assert abs(d['rot_y'] - ang[1]) < 5e-7 and abs(d['alpha'] - alpha) < 5e-7
            assert abs(d['score'] - raw.get('score', 1.0)) < 5e-9
            assert fmt.get_instance_str(d) == line                                           # idempotent
    with pytest.raises(ValueError):
        fmt.parse_label_line('Car 0 0 1.0')


@pytest.mark.gpu
def test_post_process_writes_one_file_per_image(tmp_path):
    """EgoNet.post_process(save_dict=...) -- the pose solve runs on the GPU."""
    from egonet_amd import configs
    from egonet_amd.model.egonet import EgoNet
    c = _cases()[2]
    ego = EgoNet(configs.tiny_config(), pre_trained=False).cuda()
    rng = np.random.RandomState(0)
    n = len(c['alphas'])
    rec = {'raw_txt_format': c['raw_txt_format'], 'kpts_3d_pred': rng.randn(n, 32, 3) + np.array([0, 0, 20.0]),
           'kpts_2d_pred': [rng.rand(1, 66) * 300 for _ in range(n)],
           'K': np.array([[707.0493, 0., 604.0814], [0., 707.0493, 180.5066], [0., 0., 1.]])}
    out = ego.post_process({'/data/kitti/image_2/000123.png': rec},
                           save_dict={'flag': True, 'save_dir': str(tmp_path)}, alpha_mode='proj')
    path = tmp_path / '000123.txt'
    assert path.read_text() == out['/data/kitti/image_2/000123.png']['pred_str']
    lines = path.read_text().split('\n')
    assert len(lines) == n
    got = fmt.parse_label_line(lines[0])
    r = out['/data/kitti/image_2/000123.png']
    assert abs(got['rot_y'] - r['euler_angles'][0, 1]) < 5e-7 and abs(got['alpha'] - r['alphas'][0]) < 5e-7
    # without the flag nothing is written and no string is produced
    rec2 = dict(rec)
    rec2.pop('pred_str', None)
    out2 = ego.post_process({'x/000124.png': rec2}, save_dict={'flag': False, 'save_dir': str(tmp_path)},
                            alpha_mode='trans')
    assert 'pred_str' not in out2['x/000124.png'] and not (tmp_path / '000124.txt').exists()
