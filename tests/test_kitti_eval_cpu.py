"""KITTI AP / AOS evaluator (csrc/kitti_eval.cpp, host-only C++) against the
independent Python restatement (oracle/kitti_eval_oracle.py) and hand-computed
cases.  The reference's own evaluator needs Boost and cannot be built here:
parity with its binary is unpinned."""
import math
import os

import numpy as np
import pytest

from egonet_amd import evaluate
from egonet_amd.common import format as fmt
from oracle import kitti_eval_oracle as orc


def _gt_line(cls, trunc, occ, alpha, box, ry=0.1):
    return '%s %.2f %d %.4f %.2f %.2f %.2f %.2f 1.5 1.6 3.9 1.0 1.5 20.0 %.4f' % ((cls, trunc, occ, alpha) + tuple(box) + (ry,))


def _det_line(cls, alpha, box, score):
    return fmt.get_instance_str({'class': cls, 'truncation': 0.0, 'occlusion': 0.0, 'alpha': alpha, 'bbox': list(box),
                                 'dimensions': [3.9, 1.5, 1.6], 'locations': [1.0, 1.5, 20.0], 'rot_y': 0.1,
                                 'score': score})


def _write(tmp, frames):
    gt_dir, res_dir = tmp / 'label_2', tmp / 'result'
    (res_dir / 'data').mkdir(parents=True)
    gt_dir.mkdir()
    for idx, (gts, dets) in frames.items():
        (gt_dir / ('%06d.txt' % idx)).write_text('\n'.join(gts) + ('\n' if gts else ''))
        (res_dir / 'data' / ('%06d.txt' % idx)).write_text('\n'.join(dets) + ('\n' if dets else ''))
    return str(gt_dir), str(res_dir)


def _random_frames(seed, n_frames=25):
    rng = np.random.RandomState(seed)
    frames = {}
    for f in range(n_frames):
        gts, dets = [], []
        for _ in range(rng.randint(0, 7)):
            cls = rng.choice(['Car', 'Car', 'Car', 'Van', 'Pedestrian', 'Person_sitting', 'Cyclist', 'DontCare', 'Truck'])
            x1, y1 = rng.uniform(0, 1100), rng.uniform(100, 300)
            w, h = rng.uniform(20, 200), rng.uniform(15, 120)
            box = (x1, y1, x1 + w, y1 + h)
            alpha = rng.uniform(-3.1, 3.1)
            gts.append(_gt_line(cls, rng.choice([0.0, 0.1, 0.25, 0.4, 0.7]), rng.randint(0, 4), alpha, box))
            if cls != 'DontCare' and rng.rand() < 0.8:          # a detection near most objects
                j = rng.uniform(-0.12, 0.12, 4) * np.array([w, h, w, h])
                dcls = cls if cls in ('Car', 'Pedestrian', 'Cyclist') else rng.choice(['Car', 'Pedestrian'])
                dets.append(_det_line(dcls, alpha + rng.normal(0, 0.4), np.array(box) + j, rng.uniform(0.05, 1.0)))
            elif cls == 'DontCare' and rng.rand() < 0.7:        # a detection inside a DontCare area
                dets.append(_det_line('Car', 0.3, (x1 + 1, y1 + 1, x1 + w * 0.6, y1 + max(h * 0.6, 30)), rng.uniform(0.05, 1.0)))
        for _ in range(rng.randint(0, 3)):                       # false positives, some too small
            x1, y1 = rng.uniform(0, 1100), rng.uniform(100, 300)
            dets.append(_det_line(rng.choice(['Car', 'Pedestrian', 'Cyclist']), rng.uniform(-3, 3),
                                  (x1, y1, x1 + rng.uniform(20, 150), y1 + rng.uniform(10, 90)), rng.uniform(0.05, 1.0)))
        frames[f * 3 + 1] = (gts, dets)
    return frames


def _oracle(frames):
    parsed = [orc.parse_frame(frames[k][0], frames[k][1]) for k in sorted(frames)]
    return orc.evaluate(parsed)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_cpp_evaluator_equals_python_restatement(tmp_path, seed):
    frames = _random_frames(seed)
    gt_dir, res_dir = _write(tmp_path, frames)
    got = evaluate.evaluate_aos(gt_dir, res_dir)
    want, aos_valid = _oracle(frames)
    assert got['n_frames'] == len(frames) and got['aos_valid'] == aos_valid is True
    assert set(k for k in got if k in evaluate.CLASSES) == set(want)
    for name, (prec, aos) in want.items():
        np.testing.assert_array_equal(got[name]['precision'], np.array(prec))      # same doubles, NaN == NaN
        np.testing.assert_array_equal(got[name]['aos'], np.array(aos))
        assert all(0 <= v <= 100 for v in got[name]['AP'])
    assert np.count_nonzero(got['car']['precision'][2]) >= 5       # a real precision/recall curve, not a trivial one


def test_perfect_detections_and_a_known_orientation_error(tmp_path):
    """80 easy cars (2 per frame) detected exactly with distinct scores, orientation off
    by 0.5 rad everywhere: 80 recall steps -> all 41 samples are reached, precision = 1
    and AOS = (1 + cos 0.5) / 2 at every one of them."""
    box = lambda k: (100.0 + 300 * k, 150.0, 200.0 + 300 * k, 230.0)          # noqa: E731
    frames = {}
    for f in range(40):
        frames[f] = ([_gt_line('Car', 0.0, 0, 0.2 * k, box(k)) for k in range(2)],
                     [_det_line('Car', 0.2 * k + 0.5, box(k), 0.99 - 0.01 * (2 * f + k)) for k in range(2)])
    gt_dir, res_dir = _write(tmp_path, frames)
    res = evaluate.evaluate_aos(gt_dir, res_dir)
    assert list(res) == ['n_frames', 'aos_valid', 'car']
    sim = (1 + math.cos(0.5)) / 2
    for lv in range(3):
        np.testing.assert_allclose(res['car']['precision'][lv], 1.0)
        np.testing.assert_allclose(res['car']['aos'][lv], sim, rtol=1e-12)
        assert abs(res['car']['AP'][lv] - 100.0) < 1e-9 and abs(res['car']['AOS'][lv] - 100.0 * sim) < 1e-9


def test_difficulty_filters_dontcare_and_neighbour_classes(tmp_path):
    """Hand-computed: one easy car found; one occluded car (moderate only) found; a Van
    with a 'Car' detection on it and a detection inside a DontCare area are neither TP
    nor FP; one stray detection is the only false positive."""
    frames = {1: ([_gt_line('Car', 0.0, 0, 0.0, (100, 100, 200, 180)),
                   _gt_line('Car', 0.0, 1, 0.0, (300, 100, 400, 180)),
                   _gt_line('Van', 0.0, 0, 0.0, (500, 100, 620, 200)),
                   _gt_line('DontCare', -1, -1, -10, (700, 100, 900, 250))],
                  [_det_line('Car', 0.0, (100, 100, 200, 180), 0.9),
                   _det_line('Car', 0.0, (300, 100, 400, 180), 0.8),
                   _det_line('Car', 0.0, (500, 100, 620, 200), 0.7),
                   _det_line('Car', 0.0, (720, 120, 800, 200), 0.6),
                   _det_line('Car', 0.0, (1000, 100, 1100, 190), 0.5)])}
    gt_dir, res_dir = _write(tmp_path, frames)
    res = evaluate.evaluate_aos(gt_dir, res_dir)
    easy, moderate = res['car']['precision'][0], res['car']['precision'][1]
    # the arrays are indexed by recall STEP reached (one entry per score threshold), :679-696
    # easy: 1 gt -> one threshold (0.9); the 0.8..0.5 detections are below it: precision 1
    assert easy[0] == 1.0 and not easy[1:].any()
    # moderate: 2 gts -> thresholds 0.9 and 0.8; above 0.8 there is no false positive (the Van
    # and DontCare detections are absorbed, the stray one scores 0.5): precision 1, 1
    assert moderate[0] == 1.0 and moderate[1] == 1.0 and not moderate[2:].any()
    want, _ = _oracle(frames)
    np.testing.assert_array_equal(res['car']['precision'], np.array(want['car'][0]))


def test_invalid_orientation_disables_aos_and_errors_are_reported(tmp_path):
    frames = {3: ([_gt_line('Car', 0.0, 0, 0.0, (100, 100, 200, 180))],
                  [_det_line('Car', -10, (100, 100, 200, 180), 0.9)])}
    gt_dir, res_dir = _write(tmp_path, frames)
    res = evaluate.evaluate_aos(gt_dir, res_dir)
    assert res['aos_valid'] is False and res['car']['AOS'] is None and res['car']['precision'][0][0] == 1.0
    os.remove(os.path.join(gt_dir, '000003.txt'))
    with pytest.raises(FileNotFoundError):
        evaluate.evaluate_aos(gt_dir, res_dir)
    with pytest.raises(FileNotFoundError):
        evaluate.evaluate_aos(gt_dir, str(tmp_path / 'nowhere'))
