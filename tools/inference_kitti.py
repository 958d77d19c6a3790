"""End-to-end inference on a KITTI-style directory (the flow of the reference's
``tools/inference.py:main/inference`` for BASELINE config 5): frames + 2D boxes ->
GPU crops -> HRNet key-points -> lifter -> pose solve -> KITTI result files ->
(optionally) 2D AP / AOS against labels.

    python tools/inference_kitti.py --images <dir of png> --boxes <dir of KITTI label/detection txt>
        [--calib <dir>] --out <result dir> [--ckpt <dir with HC.pth L.pth LS.npy> | --synthetic]
        [--gt <label dir>] [--classes Car] [--conf-thres 0] [--alpha-mode proj|trans] [--frames-per-step 8]

Multi-GPU (BASELINE config 5's 8-GPU form): launch one process per GPU with
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...``;
the frames are sharded contiguously by rank (egonet_amd.parallel.shard_range), every rank holds a
full weight copy and writes the result files of its own frames -- no collective on the data path;
rank 0 waits at a barrier, fills in the empty files and runs the evaluator.

Boxes come from KITTI lines (ground-truth labels = the reference's ``use_gt_box``, or a 2D/3D
detector's result files = ``use_pred_box``).  Every frame is uploaded once as uint8; all its
boxes are cropped in one launch (csrc/crop.hip).  Result files go to ``<out>/data/%06d.txt``
in the format the evaluator reads; frames without predictions get empty files
(tools/inference.py:198-210).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egonet_amd import configs, evaluate, synth                      # noqa: E402
from egonet_amd.common import crop_gpu                                # noqa: E402
from egonet_amd.common import format as kfmt                          # noqa: E402
from egonet_amd.model.egonet import EgoNet                            # noqa: E402

KITTI_K = np.array([[707.0493, 0., 604.0814], [0., 707.0493, 180.5066], [0., 0., 1.]])


def read_calib(path):
    """P2 of a KITTI calib file -> K (3x3); the default KITTI intrinsics when absent."""
    if path and os.path.isfile(path):
        with open(path) as f:
            for line in f:
                if line.startswith('P2:'):
                    p = np.array([float(v) for v in line.split()[1:13]]).reshape(3, 4)
                    return p[:, :3].copy()
    return KITTI_K.copy()


def read_boxes(path, classes, thres):
    rows = []
    if os.path.isfile(path):
        with open(path) as f:
            for line in f:
                if len(line.split()) >= 15:
                    d = kfmt.parse_label_line(line)
                    if d['class'].lower() in classes and d.get('score', 1.0) >= thres:
                        rows.append(d)
    return rows


def build_model(a):
    cfg = configs.w48_config('coordinates') if not a.tiny else configs.hrnet_config(
        8, (64, 64), 33, 'coordinates', modules=(1, 1, 1), num_blocks=1, lifter_neurons=128)
    if a.ckpt:
        cfg['dirs'] = {'ckpt': a.ckpt}
        return EgoNet(cfg, pre_trained=True).eval().cuda()
    ego = EgoNet(cfg, pre_trained=False)               # --synthetic: seeded random weights (no checkpoint offline)
    ego.HC.load_state_dict(synth.synth_state_dict(ego.HC.state_dict(), seed=6))
    ego.L.load_state_dict(synth.synth_state_dict(ego.L.state_dict(), seed=7))
    ego.LS = synth.synth_lifter_stats(66, 96, seed=1)
    return ego.eval().cuda()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', required=True)
    ap.add_argument('--boxes', required=True)
    ap.add_argument('--calib', default=None)
    ap.add_argument('--out', required=True)
    ap.add_argument('--ckpt', default=None)
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--tiny', action='store_true', help='tiny HC / lifter (tests)')
    ap.add_argument('--gt', default=None)
    ap.add_argument('--classes', default='Car')
    ap.add_argument('--conf-thres', type=float, default=0.0)
    ap.add_argument('--alpha-mode', default='proj', choices=['proj', 'trans'])
    ap.add_argument('--frames-per-step', type=int, default=8)
    a = ap.parse_args(argv)
    if not a.ckpt and not a.synthetic:
        ap.error('give --ckpt <dir> or --synthetic')
    classes = {c.strip().lower() for c in a.classes.split(',')}
    data_dir = os.path.join(a.out, 'data')
    os.makedirs(data_dir, exist_ok=True)
    # one process per GPU: contiguous shard of the frame list per rank, no data-path collective
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        from egonet_amd.parallel import shard_range
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    ego = build_model(a)
    names = sorted(f for f in os.listdir(a.images) if f.lower().endswith(('.png', '.jpg', '.jpeg')))
    all_names = names
    if world > 1:
        lo_r, hi_r = shard_range(len(names), world, rank)
        names = names[lo_r:hi_r]
    n_inst, t0 = 0, time.perf_counter()
    for lo in range(0, len(names), a.frames_per_step):
        annot = {'path': [], 'boxes': [], 'raw_txt_format': [], 'K': []}
        images = {}
        for name in names[lo:lo + a.frames_per_step]:
            stem = os.path.splitext(name)[0]
            rows = read_boxes(os.path.join(a.boxes, stem + '.txt'), classes, a.conf_thres)
            if not rows:
                continue
            path = os.path.join(a.images, name)
            annot['path'].append(path)
            annot['boxes'].append(np.array([r['bbox'] for r in rows], dtype=np.float64))
            annot['raw_txt_format'].append(rows)
            annot['K'].append(read_calib(os.path.join(a.calib, stem + '.txt') if a.calib else None))
            images[path] = crop_gpu.load_rgb(path)
        if not annot['path']:
            continue
        records = ego(annot, images=images)
        ego.post_process(records, save_dict={'flag': True, 'save_dir': data_dir}, alpha_mode=a.alpha_mode)
        n_inst += sum(len(b) for b in annot['boxes'])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        cnt = torch.tensor([float(n_inst), dt], dtype=torch.float64, device='cuda')
        tmax = cnt[1:].clone()
        dist.all_reduce(cnt[:1])                         # instances of the whole job
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)      # slowest rank (also the barrier before the evaluator)
        n_inst, dt = int(cnt[0].item()), float(tmax.item())
        if rank != 0:
            dist.destroy_process_group()
            return None
    names = all_names
    written = set(os.listdir(data_dir))
    for name in names:                                   # frames without predictions: empty result files
        stem = os.path.splitext(name)[0] + '.txt'
        if stem not in written:
            open(os.path.join(data_dir, stem), 'w').close()
    out = {'frames': len(names), 'instances': n_inst, 'seconds': round(dt, 3), 'n_gpus': world,
           'instances_per_s': round(n_inst / dt, 1) if dt > 0 else None, 'result_dir': data_dir}
    if dist is not None:
        dist.destroy_process_group()
    if a.gt:
        res = evaluate.evaluate_aos(a.gt, a.out)
        out['eval'] = {k: {'AP': v['AP'], 'AOS': v['AOS']} for k, v in res.items() if isinstance(v, dict)}
    print(json.dumps(out))
    return out


if __name__ == '__main__':
    main()
