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

What the wrapper guarantees:
  * the warm-up iterations leave NO trace: parameters, Adam moments, the step counter and every
    BatchNorm buffer (running statistics, ``num_batches_tracked``) are restored before the capture,
    so the first replay is iteration 1 of the optimisation;
  * ``trainer.lr`` changes (``MultiStepLR`` in ``trainer.train``) reach the replayed graph: the
    learning rate lives in device memory and ``__call__`` refreshes it before the replay;
  * the trainer's side stream for weight gradients is switched off while the graph owns the
    trainer and handed back by ``close()`` (or when the wrapper is dropped).
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
        self._saved_stream = getattr(trainer, 'wgrad_stream', None)
        if self._saved_stream is not None:
            trainer.wgrad_stream = None
            if hasattr(trainer, '_wgrad_ws'):
                trainer._wgrad_ws = None
        self.static = [None if t is None else t.clone() for t in inputs]
        self.kwargs = kwargs
        dev = next(t for t in self.static if t is not None).device
        flat = trainer.flat
        with torch.cuda.device(dev):
            # everything an iteration mutates, to undo the warm-up
            state = [flat.flat, flat.m, flat.v, flat.step_dev] + list(trainer.model.buffers())
            snap = [t.clone() for t in state]
            # warm up on a side stream (allocations, tile tuning, lazy inits), then capture
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    trainer.step(*self.static, **kwargs)
                with torch.no_grad():
                    for t, s0 in zip(state, snap):
                        t.copy_(s0)
            torch.cuda.current_stream().wait_stream(side)
            del snap
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = trainer.step(*self.static, **kwargs)
        self._invalidate()

    def _invalidate(self):
        from .engine import invalidate
        invalidate(self.trainer.model)      # replays write the weights through raw pointers

    def __call__(self, *inputs):
        flat = self.trainer.flat
        if self.trainer.lr != flat._lr_host:          # scheduler step since the last iteration
            flat.lr_dev.fill_(self.trainer.lr)
            flat._lr_host = self.trainer.lr
        for dst, src in zip(self.static, inputs):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self._invalidate()
        return self.loss

    def close(self):
        """Give the trainer its weight-gradient side stream back (eager steps after the graph)."""
        if self._saved_stream is not None and self.trainer is not None:
            self.trainer.wgrad_stream = self._saved_stream
            self._saved_stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
