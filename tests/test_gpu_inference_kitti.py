"""BASELINE config 5 end to end on a synthetic KITTI-style directory
(tools/inference_kitti.py): PNG frames + label boxes -> GPU crops -> key-points -> lifter
-> pose solve -> result files -> AP / AOS evaluator."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))

from egonet_amd.common import format as kfmt        # noqa: E402

pytestmark = pytest.mark.gpu


def _label(cls, alpha, box, occ=0):
    return '%s 0.00 %d %.4f %.2f %.2f %.2f %.2f 1.50 1.60 3.90 1.00 1.50 20.00 %.4f' % ((cls, occ, alpha) + box + (alpha + 0.05,))


def test_frames_to_result_files_to_aos(tmp_path, monkeypatch):
    from PIL import Image
    import inference_kitti
    monkeypatch.setenv('EGONET_AMD_AUTOTUNE', '0')
    img_dir, lab_dir, cal_dir, out_dir = (tmp_path / n for n in ('image_2', 'label_2', 'calib', 'out'))
    for d in (img_dir, lab_dir, cal_dir):
        d.mkdir()
    rng = np.random.RandomState(0)
    labels = {
        0: [_label('Car', 0.3, (100.0, 150.0, 260.0, 250.0)), _label('Car', -1.2, (600.0, 160.0, 700.0, 230.0)),
            _label('Pedestrian', 0.1, (400.0, 150.0, 430.0, 230.0))],
        1: [_label('Pedestrian', 0.5, (300.0, 150.0, 330.0, 240.0)), _label('DontCare', -10, (0.0, 0.0, 50.0, 50.0))],
        2: [_label('Car', 2.0, (900.0, 170.0, 1100.0, 300.0)), _label('Van', 0.0, (20.0, 180.0, 120.0, 260.0))],
    }
    for idx, lines in labels.items():
        Image.fromarray(rng.randint(0, 256, (375, 1242, 3)).astype(np.uint8)).save(str(img_dir / ('%06d.png' % idx)))
        (lab_dir / ('%06d.txt' % idx)).write_text('\n'.join(lines) + '\n')
        (cal_dir / ('%06d.txt' % idx)).write_text(
            'P0: ' + ' '.join(['0'] * 12) + '\nP2: 721.5377 0 609.5593 44.85728 0 721.5377 172.854 0.2163791 0 0 1 0.002745884\n')
    out = inference_kitti.main(['--images', str(img_dir), '--boxes', str(lab_dir), '--calib', str(cal_dir),
                                '--out', str(out_dir), '--synthetic', '--tiny', '--gt', str(lab_dir),
                                '--frames-per-step', '2'])
    assert out['frames'] == 3 and out['instances'] == 3
    files = sorted(os.listdir(str(out_dir / 'data')))
    assert files == ['000000.txt', '000001.txt', '000002.txt']
    assert (out_dir / 'data' / '000001.txt').read_text() == ''                 # no car: empty file
    lines0 = (out_dir / 'data' / '000000.txt').read_text().split('\n')
    assert len(lines0) == 2
    for line, src in zip(lines0, labels[0][:2]):
        got, want = kfmt.parse_label_line(line), kfmt.parse_label_line(src)
        assert got['class'] == 'Car' and got['bbox'] == want['bbox']
        assert got['dimensions'] == want['dimensions'] and got['locations'] == want['locations']
        assert -math.pi <= got['alpha'] <= math.pi and math.isfinite(got['rot_y'])
        assert (got['alpha'], got['rot_y']) != (want['alpha'], want['rot_y'])   # replaced by the predictions
        assert got['score'] == 1.0
    ev = out['eval']['car']
    # the boxes are the labels' own: every evaluated car is found, nothing else is a false positive
    assert all(0.0 < v <= 100.0 for v in ev['AP']) and all(0.0 <= v <= 100.0 for v in ev['AOS'])
    assert ev['AOS'][0] <= ev['AP'][0] + 1e-9                                   # orientation similarity <= 1
