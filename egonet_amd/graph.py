"""A whole native training iteration as ONE hipGraph launch.

The native steps (``train_hrnet.HRNetTrainStep``, ``train_lifter.LifterTrainStep``)
issue a few hundred to a few thousand HIP launches per iteration from Python.
Shapes are static, the Adam step counter and the learning rate live in device
memory (``egn_adam_step_dev_f32``), dropout masks come from torch's
capture-aware Philox generator -- so the launch list of one iteration can be
captured once (``hipStreamBeginCapture`` via ``torch.cuda.graph``) and replayed:
the host cost of an iteration becomes one ``hipGraphLaunch``.  This is what
keeps small per-GPU batches (the reference trains HC with 24 crops over its
GPUs, KITTI_train_IGRs.yml) from being launch-bound.

    g = GraphedStep(step, images, target, joints)      # device tensors: they become the static inputs
    loss = g(images2, target2, joints2)                # copies into the static inputs, replays

Gradient all-reduce (``grad_sync``) is not captured: use one process per GPU
with eager steps, or capture with ``grad_sync=None`` on a single GPU.
"""
import torch


class GraphedStep(object):
    def __init__(self, trainer, *inputs, warmup=2, **kwargs):
        if getattr(trainer, 'grad_sync', None) is not None:
            raise NotImplementedError('GraphedStep does not capture the gradient all-reduce')
        for t in inputs:
            if t is not None and not (torch.is_tensor(t) and t.is_cuda):
                raise ValueError('GraphedStep inputs must be CUDA tensors (they become the static graph inputs)')
        self.trainer = trainer
        # the trainer's side stream for weight gradients helps eager steps (-4 %); inside a graph the
        # forked branch replays slower than one chain (78.4 vs 76 ms measured), so capture one stream
        if getattr(trainer, 'wgrad_stream', None) is not None:
            trainer.wgrad_stream = None
            trainer._wgrad_ws = None
        self.static = [None if t is None else t.clone() for t in inputs]
        self.kwargs = kwargs
        dev = next(t for t in self.static if t is not None).device
        with torch.cuda.device(dev):
            # warm up on a side stream (allocations, tile tuning, cuBLAS-style lazy inits), then capture
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    trainer.step(*self.static, **kwargs)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = trainer.step(*self.static, **kwargs)

    def __call__(self, *inputs):
        for dst, src in zip(self.static, inputs):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss
