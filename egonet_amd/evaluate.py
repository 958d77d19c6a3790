"""KITTI 2D AP / AOS evaluation (csrc/kitti_eval.cpp; the IMAGE-metric path of the
reference's ``tools/kitti-eval/evaluate_object_3d_offline.cpp`` without Boost).

    res = evaluate_aos(gt_dir, result_dir)      # result_dir/data/%06d.txt as written by
    res['car']['AOS']                           # EgoNet.post_process(save_dict=...)
    -> [easy, moderate, hard] in percent (11-point summary, evaluate...cpp:720-724)
"""
import ctypes as C

import numpy as np

from . import _lib

CLASSES = ('car', 'pedestrian', 'cyclist')
LEVELS = ('easy', 'moderate', 'hard')


def evaluate_aos(gt_dir, result_dir):
    L = _lib.lib()
    prec = np.zeros((3, 3, 41), dtype=np.float64)
    aos = np.zeros((3, 3, 41), dtype=np.float64)
    evaluated = (C.c_int * 3)()
    n_frames, aos_valid = C.c_int(0), C.c_int(0)
    rc = L.egn_kitti_eval_image(str(gt_dir).encode(), str(result_dir).encode(), C.byref(n_frames), evaluated,
                                C.byref(aos_valid), prec.ctypes.data_as(C.POINTER(C.c_double)),
                                aos.ctypes.data_as(C.POINTER(C.c_double)))
    if rc == -2:
        raise FileNotFoundError('a result file has no ground-truth file in %s' % gt_dir)
    if rc == -3:
        raise FileNotFoundError('no result files under %s/data' % result_dir)
    if rc != 0:
        raise ValueError('egn_kitti_eval_image: code %d' % rc)
    out = {'n_frames': n_frames.value, 'aos_valid': bool(aos_valid.value)}
    for c, name in enumerate(CLASSES):
        if not evaluated[c]:
            continue
        out[name] = {'precision': prec[c].copy(), 'aos': aos[c].copy() if aos_valid.value else None,
                     'AP': [float(prec[c, l, ::4].sum() / 11 * 100) for l in range(3)],
                     'AOS': [float(aos[c, l, ::4].sum() / 11 * 100) for l in range(3)] if aos_valid.value else None}
    return out
