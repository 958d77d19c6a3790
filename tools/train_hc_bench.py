"""BASELINE config 4: HRNet-W48 heat-map/coordinate model, forward + backward +
Adam on synthetic crops, native HIP path (egonet_amd.train_hrnet), one process
per GPU with an RCCL all-reduce of the flat gradient when WORLD_SIZE > 1.

    python tools/train_hc_bench.py [--batch 32] [--steps 5] [--warmup 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 tools/train_hc_bench.py --batch 32          # 256 crops / step

Prints one JSON line on rank 0: crops/s (whole job), ms/step, algorithmic
TFLOP/s (3 x 42.035 GFLOP per crop).  Data: synthetic crops,
uniform random target maps / joints, seeded synthetic weights.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egonet_amd import configs, parallel, synth                        # noqa: E402
from egonet_amd.model.heatmapModel import hrnet                         # noqa: E402
from egonet_amd.train_hrnet import HRNetTrainStep                       # noqa: E402

GFLOP_FWD_PER_CROP = 42.035


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32, help='crops per GPU')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--graph', action='store_true', help='replay the iteration as one hipGraph (single GPU)')
    ap.add_argument('--main-priority', type=int, default=None,
                    help='run the step on a stream of this HIP priority (-1 = high) instead of the default stream')
    a = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group('nccl', rank=rank, world_size=world)
    cfg = configs.w48_config('coordinates')
    net = hrnet.get_pose_net(cfg, is_train=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=1))
    net = net.cuda().train()
    parallel.broadcast_module(net, src=0)
    tr = HRNetTrainStep(net, lr=1e-3, grad_sync=parallel.FlatGradSync(32.0) if world > 1 else None)
    g = torch.Generator().manual_seed(100 + rank)
    x = synth.synth_crops(a.batch, 3, 256, 256, seed=50 + rank).cuda()
    tgt = torch.rand(a.batch, 33, 64, 64, generator=g).cuda()
    jt = (torch.rand(a.batch, 33, 2, generator=g) * 256).cuda()
    if a.main_priority is not None:
        print('stream priority range (least, greatest):', torch.cuda.Stream.priority_range(), file=sys.stderr)
        main_stream = torch.cuda.Stream(priority=a.main_priority)
        main_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(main_stream)
    for _ in range(a.warmup):
        tr.step(x, tgt, jt)
    step = lambda: tr.step(x, tgt, jt)                     # noqa: E731
    if a.graph:
        from egonet_amd.graph import GraphedStep
        graphed = GraphedStep(tr, x, tgt, jt, warmup=1)
        step = lambda: graphed(x, tgt, jt)                 # noqa: E731
        step()

    import gc
    gc.collect()
    gc.freeze()             # as egonet_amd.trainer.train does after its first iteration

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = torch.tensor([time.perf_counter() - t0], device='cuda')
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    sec = float(dt.item()) / a.steps
    if rank == 0:
        crops = a.batch * world
        print(json.dumps({
            'metric': 'hc_train_crops_per_sec', 'value': round(crops / sec, 2), 'unit': 'crops/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(sec * 1e3, 2), 'higher_is_better': True,
            'scaling': 'weak', 'dtype': 'f32', 'data': 'synthetic',
            'algorithmic_tflops_per_gpu': round(3 * GFLOP_FWD_PER_CROP * a.batch / sec / 1e3, 2),
            'loss': float(loss.item()),
            'config': {'workload': 'train_IGRs HRNet-W48 256x256 fwd+bwd+Adam, JointsCompositeLoss(mse,l1)',
                       'batch_per_gpu': a.batch, 'hipgraph': bool(a.graph), 'global_batch': crops, 'parallelism': 'dp%d' % world},
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
