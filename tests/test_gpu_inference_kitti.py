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
    # exact values: the independent Python restatement of the evaluator (oracle/kitti_eval_oracle.py, the
    # checker) on the files this run wrote gives the same doubles as the C++ evaluator
    from oracle import kitti_eval_oracle as orc
    frames = []
    for idx in sorted(labels):
        det = (out_dir / 'data' / ('%06d.txt' % idx)).read_text().split('\n')
        frames.append(orc.parse_frame(labels[idx], [l for l in det if l.strip()]))
    res, aos_valid = orc.evaluate(frames)
    assert aos_valid and set(res) == {'car'}
    prec, aos = res['car']
    want_ap = [sum(prec[l][::4]) / 11 * 100 for l in range(3)]
    want_aos = [sum(aos[l][::4]) / 11 * 100 for l in range(3)]
    assert ev['AP'] == want_ap and ev['AOS'] == want_aos
    # the boxes are the labels' own: every evaluated car is found, nothing else is a false positive.
    # Three detections with equal scores give three thresholds, i.e. precision 1 at recall samples
    # 0..2 only; of the 11 summary points (every 4th sample) just the first is non-zero.
    assert all(abs(v - 100.0 / 11) < 1e-12 for v in ev['AP'])
    # ... and AOS there is the mean orientation similarity (1 + cos(alpha_pred - alpha_gt)) / 2 of the
    # three cars (evaluate_object_3d_offline.cpp:547-548)
    sims = []
    for idx in (0, 2):
        det = [kfmt.parse_label_line(l) for l in (out_dir / 'data' / ('%06d.txt' % idx)).read_text().split('\n') if l.strip()]
        gts = [kfmt.parse_label_line(l) for l in labels[idx] if l.startswith('Car')]
        sims += [(1 + math.cos(d['alpha'] - g['alpha'])) / 2 for d, g in zip(det, gts)]
    assert len(sims) == 3
    assert abs(ev['AOS'][0] - 100.0 / 11 * sum(sims) / 3) < 1e-9
