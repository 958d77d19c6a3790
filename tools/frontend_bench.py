"""PCIe-inclusive rates of the inference path, B crops per step:

  resident   crops already in HBM (what bench.py reports as `value`)
  host_fp32  the reference's hand-over: fp32 crops produced on the host (cv2 warp +
             ToTensor + Normalize), copied host -> device every step (786 KB per crop)
  frames_u8  uint8 KITTI-sized frames copied host -> device, crops cut on the GPU
             (csrc/crop.hip), 8 boxes per frame
  *_pipelined [round 6] the same two hand-overs as a serving loop would run them: two batches in flight (engine
             slots 0 / 1 on two compute streams), the uploads of batch i + 1 on a copy stream under batch i's kernels,
             results (< 1 KB per crop) copied back asynchronously into pinned memory and consumed one step later;
             bit-identical to the one-batch-at-a-time form (asserted here).
Every row ends with the results ON THE HOST (tools/inference.py:135-199 writes them to files).

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

    # ---- two batches in flight: uploads on a copy stream, results back through pinned memory ----------------------
    from egonet_amd.common.crop_gpu import forward_affines
    copy_s = torch.cuda.Stream()
    comp = [torch.cuda.Stream(), torch.cuda.Stream()]
    aff_host = torch.from_numpy(forward_affines(centers, scales, (256, 256))).pin_memory()
    c_dev, s_dev = torch.from_numpy(centers).cuda(), torch.from_numpy(scales).cuda()     # (box geometry of the batch: 2 KB)
    slots = []
    for k in range(2):
        slots.append(dict(frames=[torch.empty(375, 1242, 3, dtype=torch.uint8, device='cuda') for _ in range(nframes)],
                          crops=torch.empty(B, 3, 256, 256, device='cuda'), aff=torch.empty(B, 6, dtype=torch.float64, device='cuda'),
                          ready=torch.cuda.Event(), free=torch.cuda.Event(), done=torch.cuda.Event(), host=None, dev=None))
    state = {'i': 0, 'last': None}

    def _consume(sl):
        sl['done'].synchronize()
        return {k: v.numpy() for k, v in sl['host'].items()}

    def _pipelined(upload_crops):
        i = state['i']
        state['i'] += 1
        sl = slots[i & 1]
        with torch.cuda.stream(copy_s):
            copy_s.wait_event(sl['free'])                 # the batch that last read this slot's inputs has consumed them
            if upload_crops:
                sl['crops'].copy_(host_crops, non_blocking=True)
            else:
                for f in range(nframes):
                    sl['frames'][f].copy_(frames[f], non_blocking=True)
                sl['aff'].copy_(aff_host, non_blocking=True)
            sl['ready'].record(copy_s)
        with torch.cuda.stream(comp[i & 1]):
            st = comp[i & 1]
            st.wait_event(sl['ready'])
            if not upload_crops:
                for f in range(nframes):
                    lo, hi = f * per, min(B, (f + 1) * per)
                    crop_gpu.crop_boxes(sl['frames'][f], None, None, (256, 256), affines=sl['aff'][lo:hi],
                                        out=sl['crops'][lo:hi])
            res = ego.infer_crops(sl['crops'], c_dev, s_dev, K=K, to_host=False, slot=i & 1)
            sl['free'].record(st)
            if sl['host'] is None:
                sl['host'] = {k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in res.items()}
            for k, v in res.items():
                sl['host'][k].copy_(v, non_blocking=True)
            sl['dev'] = res                                # (keeps the device results alive until the copies ran)
            sl['done'].record(st)
        prev = state['last']
        state['last'] = sl
        return _consume(prev) if prev is not None else None   # the host works on batch i - 1 while batch i runs

    def host_fp32_pipelined():
        return _pipelined(True)

    def frames_u8_pipelined():
        return _pipelined(False)

    def _drain():
        r = _consume(state['last'])
        state['last'] = None
        torch.cuda.synchronize()
        return r

    # same bits as one batch at a time
    for plain, piped in ((host_fp32, host_fp32_pipelined), (frames_u8, frames_u8_pipelined)):
        want = plain()
        piped(); piped()
        got = _drain()
        for k in want:
            assert np.array_equal(want[k], got[k]), k
    out = {'batch': B, 'steps': a.steps, 'boxes_per_frame': per,
           'h2d_bytes_per_step': {'host_fp32': int(host_crops.numel() * 4), 'frames_u8': int(nframes * 375 * 1242 * 3)}}
    for name, fn in (('resident', resident), ('host_fp32', host_fp32), ('frames_u8', frames_u8),
                     ('host_fp32_pipelined', host_fp32_pipelined), ('frames_u8_pipelined', frames_u8_pipelined)):
        piped = name.endswith('_pipelined')
        for _ in range(a.warmup):
            fn()
        if piped:
            _drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        if piped:
            _drain()                                       # the last batch's results are on the host too
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        out[name] = {'crops_per_s': round(B / dt, 1), 'ms_per_step': round(dt * 1e3, 3)}
    out['pipelined_over_resident'] = {k: round(out[k + '_pipelined']['crops_per_s'] / out['resident']['crops_per_s'], 4)
                                      for k in ('host_fp32', 'frames_u8')}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
