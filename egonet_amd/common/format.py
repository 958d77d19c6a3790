"""KITTI-style result lines (the reference's ``libs/common/format.py``).

``get_pred_str(record)`` / ``save_txt_file(path, record, params)`` keep the
reference's names and byte-for-byte output (format.py:24-73; line layout
``class trunc occ alpha x1 y1 x2 y2 h w l x y z ry score`` with a trailing
blank, as the KITTI evaluator reads it, evaluate_object_3d_offline.cpp:131-176).
``parse_label_line`` is the inverse (what ``raw_txt_format`` holds,
car_instance.py:448).  Host code: a few hundred bytes per instance.
"""
import os

# dimensions are stored (l, h, w) in raw_txt_format and written h w l (format.py:35)
_LINE = ('{cls} {trunc:.1f} {occ:.1f} {alpha:.6f} {b0:.6f} {b1:.6f} {b2:.6f} {b3:.6f} '
         '{h:.6f} {w:.6f} {l:.6f} {x:.6f} {y:.6f} {z:.6f} {ry:.6f} {score:.8f} ')


def get_instance_str(dic):
    dims, loc, box = dic['dimensions'], dic['locations'], dic['bbox']
    return _LINE.format(cls=dic['class'], trunc=dic['truncation'], occ=dic['occlusion'], alpha=dic['alpha'],
                        b0=box[0], b1=box[1], b2=box[2], b3=box[3], h=dims[1], w=dims[2], l=dims[0],
                        x=loc[0], y=loc[1], z=loc[2], ry=dic['rot_y'], score=dic.get('score', 1.0))


def get_pred_str(record):
    """One line per instance of ``record['raw_txt_format']`` with ``rot_y`` / ``alpha``
    replaced by the predicted yaw (``euler_angles[:, 1]``) and observation angle
    (format.py:43-60); lines joined by newlines, none after the last."""
    angles, alphas = record['euler_angles'], record['alphas']
    lines = []
    for i in range(len(angles)):
        inst = dict(record['raw_txt_format'][i])
        inst['rot_y'] = angles[i, 1]
        inst['alpha'] = alphas[i]
        lines.append(get_instance_str(inst))
    return '\n'.join(lines)


def save_txt_file(img_path, prediction, params):
    """Write ``prediction['pred_str']`` to ``<save_dir>/<image stem>.txt`` (format.py:62-73)."""
    if not params['flag']:
        return None
    save_path = os.path.join(params['save_dir'], img_path.split('/')[-1][:-3] + 'txt')
    with open(save_path, 'w') as f:
        f.write(prediction['pred_str'])
    return save_path


def parse_label_line(line):
    """A KITTI label / result line -> the dict ``raw_txt_format`` holds."""
    t = line.split()
    if len(t) < 15:
        raise ValueError('KITTI label line needs at least 15 fields, got %d' % len(t))
    v = [float(s) for s in t[1:]]
    out = {'class': t[0], 'truncation': v[0], 'occlusion': v[1], 'alpha': v[2], 'bbox': v[3:7],
           'dimensions': [v[9], v[7], v[8]], 'locations': v[10:13], 'rot_y': v[13]}
    if len(v) > 14:
        out['score'] = v[14]
    return out
