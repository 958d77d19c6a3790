"""CPU oracle: KITTI 2D AP / AOS (TEST INFRASTRUCTURE ONLY).  **PARITY UNPINNED.**

An independent Python restatement of the IMAGE-metric path of
tools/kitti-eval/evaluate_object_3d_offline.cpp (functions cited inline).  The
reference C++ needs Boost headers that the image does not have, so neither this
oracle nor csrc/kitti_eval.cpp can be run against the reference binary; the two
restatements check each other (tests/test_kitti_eval_cpu.py) together with
hand-computed cases.

Input: lists of frames; a frame = (gt rows, detection rows), rows being the
whitespace-split fields of KITTI label / result lines.
"""
import math

NAMES = ('car', 'pedestrian', 'cyclist')
MIN_HEIGHT = (40, 25, 25)
MAX_OCC = (0, 1, 2)
MAX_TRUNC = (0.15, 0.3, 0.5)
MIN_OVERLAP = (0.7, 0.5, 0.5)            # :54, the IMAGE row
SAMPLES = 41


def _box(f):
    return dict(type=f[0], alpha=float(f[3]), x1=float(f[4]), y1=float(f[5]), x2=float(f[6]), y2=float(f[7]))


def parse_frame(gt_lines, det_lines):
    gts, dets = [], []
    for ln in gt_lines:
        f = ln.split()
        if len(f) >= 15:
            g = _box(f)
            g.update(trunc=float(f[1]), occ=int(float(f[2])))
            gts.append(g)
    for ln in det_lines:
        f = ln.split()
        if len(f) >= 16:
            d = _box(f)
            d['score'] = float(f[15])
            dets.append(d)
    return gts, dets


def iou(a, b, over_a=False):                                   # :227-265
    w = min(a['x2'], b['x2']) - max(a['x1'], b['x1'])
    h = min(a['y2'], b['y2']) - max(a['y1'], b['y1'])
    if w <= 0 or h <= 0:
        return 0.0
    inter = w * h
    aa = (a['x2'] - a['x1']) * (a['y2'] - a['y1'])
    ab = (b['x2'] - b['x1']) * (b['y2'] - b['y1'])
    return inter / aa if over_a else inter / (aa + ab - inter)


def clean(cls, level, gts, dets):                              # :381-454
    name = NAMES[cls]
    gflag, dontcare, n = [], [], 0
    for g in gts:
        t = g['type'].lower()
        if t == name:
            kind = 1
        elif (name == 'pedestrian' and t == 'person_sitting') or (name == 'car' and t == 'van'):
            kind = 0
        else:
            kind = -1
        hard = g['occ'] > MAX_OCC[level] or g['trunc'] > MAX_TRUNC[level] or (g['y2'] - g['y1']) < MIN_HEIGHT[level]
        if kind == 1 and not hard:
            gflag.append(0)
            n += 1
        elif kind == 0 or (hard and kind == 1):
            gflag.append(1)
        else:
            gflag.append(-1)
        if t == 'dontcare':
            dontcare.append(g)
    dflag = []
    for d in dets:
        if int(abs(d['y1'] - d['y2'])) < MIN_HEIGHT[level]:
            dflag.append(1)
        else:
            dflag.append(0 if d['type'].lower() == name else -1)
    return gflag, dflag, dontcare, n


def stats(cls, gts, dets, dontcare, gflag, dflag, with_fp, with_aos, thresh):      # :456-615
    NONE = -10000000
    taken = [False] * len(dets)
    low = [with_fp and d['score'] < thresh for d in dets]
    tp = fp = fn = 0
    scores, deltas = [], []
    for i, g in enumerate(gts):
        if gflag[i] == -1:
            continue
        pick, valid, best, small = -1, NONE, 0.0, False
        for j, d in enumerate(dets):
            if dflag[j] == -1 or taken[j] or low[j]:
                continue
            o = iou(d, g)
            if not with_fp and o > MIN_OVERLAP[cls] and d['score'] > valid:
                pick, valid = j, d['score']
            elif with_fp and o > MIN_OVERLAP[cls] and (o > best or small) and dflag[j] == 0:
                best, pick, valid, small = o, j, 1, False
            elif with_fp and o > MIN_OVERLAP[cls] and valid == NONE and dflag[j] == 1:
                pick, valid, small = j, 1, True
        if valid == NONE and gflag[i] == 0:
            fn += 1
        elif valid != NONE and (gflag[i] == 1 or dflag[pick] == 1):
            taken[pick] = True
        elif valid != NONE:
            tp += 1
            scores.append(dets[pick]['score'])
            if with_aos:
                deltas.append(g['alpha'] - dets[pick]['alpha'])
            taken[pick] = True
    sim = 0.0
    if with_fp:
        for j in range(len(dets)):
            if not (taken[j] or dflag[j] == -1 or dflag[j] == 1 or low[j]):
                fp += 1
        stuff = 0
        for dc in dontcare:
            for j, d in enumerate(dets):
                if taken[j] or dflag[j] in (-1, 1) or low[j]:
                    continue
                if iou(d, dc, over_a=True) > MIN_OVERLAP[cls]:
                    taken[j] = True
                    stuff += 1
        fp -= stuff
        if with_aos:
            sim = sum((1.0 + math.cos(x)) / 2.0 for x in deltas) if (tp > 0 or fp > 0) else -1
    return tp, fp, fn, sim, scores


def thresholds(v, n_gt):                                       # :346-379
    v = sorted(v, reverse=True)
    t, cur = [], 0.0
    for i in range(len(v)):
        left = (i + 1) / n_gt
        right = (i + 2) / n_gt if i < len(v) - 1 else left
        if (right - cur) < (cur - left) and i < len(v) - 1:
            continue
        t.append(v[i])
        cur += 1.0 / (SAMPLES - 1.0)
    return t


def _div(a, b):
    if b == 0:
        return float('nan') if a == 0 else math.copysign(float('inf'), a)
    return a / b


def _max_from(vals, i):
    # std::max_element semantics: the first element is kept unless a later one compares greater
    best = vals[i]
    for v in vals[i + 1:]:
        if best < v:
            best = v
    return best


def eval_class(cls, level, frames, with_aos):                  # :622-706
    cleaned, allscores, n_gt = [], [], 0
    for gts, dets in frames:
        gflag, dflag, dc, n = clean(cls, level, gts, dets)
        n_gt += n
        cleaned.append((gflag, dflag, dc))
        allscores += stats(cls, gts, dets, dc, gflag, dflag, False, False, 0)[4]
    thr = thresholds(allscores, float(n_gt)) if n_gt else ([] if not allscores else None)
    acc = [[0, 0, 0, 0.0] for _ in thr]
    for (gts, dets), (gflag, dflag, dc) in zip(frames, cleaned):
        for t, th in enumerate(thr):
            tp, fp, fn, sim, _ = stats(cls, gts, dets, dc, gflag, dflag, True, with_aos, th)
            acc[t][0] += tp
            acc[t][1] += fp
            acc[t][2] += fn
            if sim != -1:
                acc[t][3] += sim
    prec, aos = [0.0] * SAMPLES, [0.0] * SAMPLES
    for i, (tp, fp, fn, sim) in enumerate(acc[:SAMPLES]):
        prec[i] = _div(tp, tp + fp)
        if with_aos:
            aos[i] = _div(sim, tp + fp)
    for i in range(min(len(acc), SAMPLES)):
        prec[i] = _max_from(prec, i)
        if with_aos:
            aos[i] = _max_from(aos, i)
    return prec, aos


def evaluate(frames):
    """frames: list of (gts, dets) from ``parse_frame``.  -> {class: (precision[3][41], aos[3][41])}, aos_valid"""
    with_aos = all(d['alpha'] != -10 for _, dets in frames for d in dets)
    out = {}
    for c, name in enumerate(NAMES):
        if any(d['type'].lower() == name and d['x1'] >= 0 for _, dets in frames for d in dets):
            res = [eval_class(c, lv, frames, with_aos) for lv in range(3)]
            out[name] = ([r[0] for r in res], [r[1] for r in res])
    return out, with_aos
