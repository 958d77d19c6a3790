"""Two ranks on one node: the native HRNet training step with the overlapped flat-gradient
all-reduce (parallel.FlatGradSync sessions: events of the backward's main and weight-gradient streams,
collectives on a communication stream).
  * over RCCL (backend ``nccl``): needs two GPUs -- the round-end test box has one, so that test skips
    there; it runs wherever ``torch.cuda.device_count() >= 2``;
  * over ``gloo`` with BOTH ranks on the one GPU (gloo reduces CUDA tensors through the host): the same
    session / stream / event logic on device tensors, runs on the 1-GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egonet_amd import configs, synth

pytestmark = pytest.mark.gpu

needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL over xGMI)')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(rank):
    g = torch.Generator().manual_seed(10 + rank)
    x = synth.synth_crops(2, 3, 64, 64, seed=40 + rank)
    return x, torch.rand(2, 5, 16, 16, generator=g), torch.rand(2, 5, 2, generator=g) * 64


def _grad_of(rank_inputs, device):
    from egonet_amd.model.heatmapModel import hrnet
    from egonet_amd.train_hrnet import HRNetTrainStep
    cfg = configs.tiny_config('coordinates')
    net = hrnet.get_pose_net(cfg, is_train=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=9))
    net = net.to(device).train()
    tr = HRNetTrainStep(net, lr=1e-3)
    x, t, j = rank_inputs
    tr.step(x.to(device), t.to(device), j, update=False)
    return tr.flat.grad.clone()


def _worker(rank, world, port, q, backend='nccl'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), EGONET_AMD_AUTOTUNE='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank if backend == 'nccl' else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from egonet_amd import parallel
    from egonet_amd.model.heatmapModel import hrnet
    from egonet_amd.train_hrnet import HRNetTrainStep
    cfg = configs.tiny_config('coordinates')
    net = hrnet.get_pose_net(cfg, is_train=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=9 + rank))     # ranks start different
    net = net.cuda().train()
    parallel.broadcast_module(net, src=0)
    tr = HRNetTrainStep(net, lr=1e-3, grad_sync=parallel.FlatGradSync(bucket_mb=0.02))
    x, t, j = _inputs(rank)
    tr.step(x.cuda(), t.cuda(), j, update=False)
    torch.cuda.synchronize()
    if rank == 0:
        q.put(tr.flat.grad.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@needs_two
def test_nccl_world2_native_step_gradient_is_the_rank_mean():
    """Per-rank BatchNorm statistics (DataParallel semantics): the all-reduced gradient equals the
    mean of the two single-process gradients, each on its own shard."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    os.environ['EGONET_AMD_AUTOTUNE'] = '0'
    want = 0.5 * (_grad_of(_inputs(0), 'cuda:0') + _grad_of(_inputs(1), 'cuda:0')).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5 * np.abs(want).max())


def test_gloo_world2_on_one_gpu_native_step_sync_session():
    """Both ranks on cuda:0, gloo backend: the session launches slices from inside the backward (after
    events of the main and the weight-gradient stream), waits for them before Adam; the reduced
    flat gradient is the mean of the two single-process gradients."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 'gloo')) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    os.environ['EGONET_AMD_AUTOTUNE'] = '0'
    want = 0.5 * (_grad_of(_inputs(0), 'cuda:0') + _grad_of(_inputs(1), 'cuda:0')).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5 * np.abs(want).max())
