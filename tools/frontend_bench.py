"""PCIe-inclusive rates of the inference path, B crops per step:

  resident   crops already in HBM (what bench.py reports as `value`)
  host_fp32  the reference's hand-over: fp32 crops produced on the host (cv2 warp +
             ToTensor + Normalize), copied host -> device every step (786 KB per crop)
  frames_u8  uint8 KITTI-sized frames copied host -> device, crops cut on the GPU
             (csrc/crop.hip), 8 boxes per frame

    python tools/frontend_bench.py [--batch 64] [--steps 20] [--warmup 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egonet_amd import configs, synth                      # noqa: E402
from egonet_amd.common import crop_gpu, img_proc            # noqa: E402
from egonet_amd.model.egonet import EgoNet                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--boxes-per-frame', type=int, default=8)
    a = ap.parse_args()
    cfg = configs.w48_config('coordinates')
    ego = EgoNet(cfg, pre_trained=False)
    ego.HC.load_state_dict(synth.synth_state_dict(ego.HC.state_dict(), seed=1))
    ego.L.load_state_dict(synth.synth_state_dict(ego.L.state_dict(), seed=2))
    ego.LS = synth.synth_lifter_stats(66, 96, seed=1)
    ego = ego.eval().cuda()
    B, per = a.batch, a.boxes_per_frame
    nframes = (B + per - 1) // per
    rng = np.random.RandomState(0)
    frames = [torch.from_numpy(rng.randint(0, 256, (375, 1242, 3)).astype(np.uint8)).pin_memory() for _ in range(nframes)]
    boxes = synth.synth_boxes(B, seed=3)
    rets = [img_proc.modify_bbox(b, 1.0) for b in boxes]
    centers, scales = np.stack([r['c'] for r in rets]), np.stack([r['s'] for r in rets])
    host_crops = synth.synth_crops(B, 3, 256, 256, seed=11).pin_memory()
    dev_crops = host_crops.cuda()
    K = np.array([[707.0493, 0., 604.0814], [0., 707.0493, 180.5066], [0., 0., 1.]])

    def resident():
        return ego.infer_crops(dev_crops, centers, scales, K=K)

    def host_fp32():
        return ego.infer_crops(host_crops.cuda(non_blocking=True), centers, scales, K=K)

    def frames_u8():
        parts = []
        for f in range(nframes):
            lo, hi = f * per, min(B, (f + 1) * per)
            parts.append(crop_gpu.crop_boxes(frames[f].cuda(non_blocking=True), centers[lo:hi], scales[lo:hi], (256, 256)))
        return ego.infer_crops(torch.cat(parts), centers, scales, K=K)

    out = {'batch': B, 'steps': a.steps, 'boxes_per_frame': per,
           'h2d_bytes_per_step': {'host_fp32': int(host_crops.numel() * 4), 'frames_u8': int(nframes * 375 * 1242 * 3)}}
    for name, fn in (('resident', resident), ('host_fp32', host_fp32), ('frames_u8', frames_u8)):
        for _ in range(a.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        out[name] = {'crops_per_s': round(B / dt, 1), 'ms_per_step': round(dt * 1e3, 3)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
