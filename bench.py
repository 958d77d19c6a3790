#!/usr/bin/env python
"""Benchmark of the EgoNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one batch of B synthetic 256x256 crops per GPU through the whole
hot path with inputs resident in HBM:
    HRNet-W48 heat-map forward -> soft-arg-max decode -> crop-to-screen affine
    -> lifter (66 -> 96) -> un-normalise -> pose solve
(BASELINE.json configs[1] "Batch=64 256x256 crops, heatmap forward +
soft-argmax", extended by the lift so that the number is the headline metric
crops/s "heatmap+decode+lift").  Crops shard across ranks with no data-path
collective (weak scaling, B per GPU fixed).

The same run then measures the two TRAINING configurations of BASELINE.json, each with its
own warm-up + timed steps bracketed by barrier + synchronize (max over ranks), and adds them to
the same JSON line (``--no-train`` skips them):
  train_hc      configs[3]: HRNet-W48 forward + backward + Adam, 32 crops/GPU, native HIP step,
                RCCL all-reduce of the flat gradient when N > 1 (whole-job crops/s)
  train_lifter  configs[2]: FC lifter forward + backward + Adam, batch 4096 (N = 1 only)
each with ms/step, algorithmic TFLOP/s, the dominant conv/GEMM kernel's roofline fraction
(hipEvents around every launch in one extra single-stream step) and the CPU training oracle's
rate on a bounded sample.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline      the dominant kernel (by time; keyed by the kernel SYMBOL that
                rocprofv3 prints) of the backbone, measured live with hipEvents
                around every launch of the native program on the stream the
                kernels run on; `traffic` = measured HBM bytes per launch from
                the committed PMC passes (profiles/), null if absent
  kernels       the same per shape class (time share, TFLOP/s, GB/s)
  cpu_baseline  the CPU oracle (same graph the reference's PyTorch-CPU path
                executes) timed on this host on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_HBM_GBPS = 8000.0
# committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/prof_r4.sh + tools/pmc_traffic.py), newest first
TRAFFIC_JSONS = [os.path.join(ROOT, 'profiles', n) for n in ('r6_pmc_traffic.json', 'r5_pmc_traffic.json', 'r4_pmc_traffic.json', 'r3_pmc_traffic.json', 'r2_pmc_traffic.json')]
WINO_EXECUTED = 16.0 / 36.0       # fused Winograd F(2x2,3x3): multiplies executed per direct-algorithm multiply
WINO4_EXECUTED = 36.0 / 144.0     # F(4x4,3x3): 36 multiplies per 16 outputs instead of 144


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--batch', type=int, default=64, help='crops per GPU per step')
    p.add_argument('--head', default='heatmap', choices=['heatmap', 'coordinates'])
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-seconds', type=float, default=15.0)
    p.add_argument('--profile-json', default='', help='write the per-op timing table here')
    p.add_argument('--no-train', action='store_true', help='skip the training blocks (configs 3 and 4)')
    p.add_argument('--train-batch', type=int, default=32, help='crops per GPU of the train_hc block')
    p.add_argument('--lifter-batch', type=int, default=4096)
    p.add_argument('--pipelined', action='store_true',
                   help='also time the K steps with TWO batches in flight on two streams (measured r4: +0.8 %%: the step is '
                        'its kernel time -- off by default)')
    p.add_argument('--live-traffic', action='store_true',
                   help='measure roofline.traffic IN THIS RUN: two short child passes of the forward under '
                        '`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (own runs, 120 s limit each; off by '
                        'default: the driver\'s bench run must not depend on the profiler)')
    p.add_argument('--traffic-child', action='store_true', help=argparse.SUPPRESS)
    p.add_argument('--dry-run', action='store_true',
                   help='launcher / rendezvous / timing-reduction plumbing only (no GPU work): what the CPU test of '
                        '`--gpus N` runs with EGONET_AMD_DIST_BACKEND=gloo')
    return p.parse_args()


def build_model(head, device):
    from egonet_amd import configs, synth
    from egonet_amd.model.egonet import EgoNet
    cfg = configs.w48_config(head)
    ego = EgoNet(cfg, pre_trained=False)
    hc_sd = synth.synth_state_dict(ego.HC.state_dict(), seed=1)
    l_sd = synth.synth_state_dict(ego.L.state_dict(), seed=2)
    ego.HC.load_state_dict(hc_sd)
    ego.L.load_state_dict(l_sd)
    ego.LS = synth.synth_lifter_stats(66, 96, seed=1)
    return cfg, ego.eval().to(device), hc_sd, l_sd


def _symbol(cfg, cout=None):
    """Kernel symbol of a tile configuration as rocprofv3 prints it (egn_conv_config_name); the 8-wave
    Winograd kernel is built per co-tile width: NT = 3 (Cout % 48 == 0) or 2."""
    from egonet_amd import _lib
    if cfg and cfg < 0:        # the dense GEMM kernels of csrc/gemm.hip (forms NT / NN / TN), not a conv tile config
        return 'gemm_kernel<%s> (csrc/gemm.hip)' % {-1: 'NT', -2: 'NN', -3: 'TN'}.get(cfg, '?')
    buf = C.create_string_buffer(128)
    if cfg and _lib.lib().egn_conv_config_name(cfg, buf, 128) == 0:
        sym = buf.value.decode()
        if cout is not None and cout % 48 and 'conv_wino8_kernel' in sym:
            sym = sym.replace(', 3>(', ', 2>(')
        if cout is not None and cout % 48 and 'conv_wino9_kernel' in sym:
            sym = sym.replace(', 3, 0, ', ', 2, 0, ')
        return sym
    return None


def _klass_cout(klass):
    m = re.search(r'->(\d+)@', klass or '')
    return int(m.group(1)) if m else None


def _is_wino(cfg):
    from egonet_amd import _lib
    return bool(cfg) and cfg > 0 and _lib.lib().egn_conv_config_kind(cfg) in (1, 2, 3)


def _executed(cfg):
    """Executed multiplies per direct-algorithm multiply of a tile configuration (1 for the direct kernels)."""
    from egonet_amd import _lib
    kind = _lib.lib().egn_conv_config_kind(cfg) if cfg and cfg > 0 else 0
    return WINO_EXECUTED if kind == 1 else (WINO4_EXECUTED if kind in (2, 3) else 1.0)


def kernel_tables(prog, ms):
    """Aggregate per-op hipEvent durations (a) by shape class, (b) by kernel symbol.  `flops` are the
    ALGORITHMIC (direct-convolution, 2*MAC) flops of SURVEY 8(d); `xflops` the flops the kernel EXECUTES on
    the matrix pipe -- 16/36 of them for the fused Winograd F(2x2,3x3) kernels."""
    by_class, by_symbol = {}, {}
    for meta, t in zip(prog.meta, ms):
        if meta['kind'] in ('fork', 'join'):
            continue
        cfg = meta.get('cfg', 0)
        if meta['kind'] == 'conv':
            sym = _symbol(cfg, _klass_cout(meta.get('klass')))
        elif meta['kind'] == 'pwpair':      # csrc/conv_pw.hip: layer1's 1x1 pair (fused) / its one-product form
            sym = 'void conv_pw_kernel<%s>(PwArgs)' % ('true' if '->64@' in meta['klass'] else 'false')
        else:
            sym = meta['kind'] + '_kernel'
        ex = meta['flops'] * (_executed(cfg) if meta['kind'] == 'conv' else 1.0)
        for table, key in ((by_class, meta['klass']), (by_symbol, sym)):
            a = table.setdefault(key, dict(name=key, kind=meta['kind'], launches=0, ms=0.0, flops=0.0, xflops=0.0,
                                           bytes=0.0))
            a['launches'] += 1
            a['ms'] += float(t)
            a['flops'] += meta['flops']
            a['xflops'] += ex
            a['bytes'] += meta['bytes']
    out = []
    for table in (by_class, by_symbol):
        rows = sorted(table.values(), key=lambda a: -a['ms'])
        total = sum(a['ms'] for a in rows)
        for a in rows:
            a['share'] = a['ms'] / total if total else 0.0
            a['tflops'] = a['flops'] / (a['ms'] * 1e-3) / 1e12 if a['ms'] else 0.0
            a['executed_tflops'] = a['xflops'] / (a['ms'] * 1e-3) / 1e12 if a['ms'] else 0.0
            a['gbps'] = a['bytes'] / (a['ms'] * 1e-3) / 1e9 if a['ms'] else 0.0
            a['avg_us'] = a['ms'] * 1e3 / a['launches']
        out.append((rows, total))
    return out


def measured_traffic(symbol):
    """(HBM bytes per launch of `symbol`, source file) from the committed PMC passes (profiles/README.md:
    FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs of bench.py / tools/train_hc_bench.py /
    tools/train_bench.py, FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes)."""
    for path in TRAFFIC_JSONS:
        try:
            with open(path) as f:
                t = json.load(f)
            ks = t['kernels']
            if symbol.startswith('gemm_kernel<'):
                # csrc/gemm.hip instantiations as rocprofv3 prints them: <A K-contiguous, B K-contiguous, ...>
                tag = {'NT': 'gemm_kernel<true, true,', 'NN': 'gemm_kernel<true, false,',
                       'TN': 'gemm_kernel<false, false,'}[symbol[12:14]]
                symbol = next(k for k in ks if tag in k)
            return ks[symbol]['hbm_bytes_per_launch'], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError, StopIteration):
            continue
    return None, None


def mfma_roofline(symbol, direct_tflops, executed_tflops):
    """The `roofline` object of an MFMA-bound kernel symbol.  `achieved` / `frac` price the flops the kernel
    EXECUTES on the fp32 matrix pipe against its dense peak (always <= 1: a roofline fraction); for the fused
    Winograd kernels, which execute 16 of every 36 direct-algorithm multiplies, the algorithmic
    (direct-convolution) rate is given beside it as `direct_equivalent_*`."""
    roof = {'bound': 'mfma', 'kernel': symbol, 'achieved': executed_tflops, 'peak': PEAK_FP32_MFMA_TFLOPS,
            'unit': 'TFLOP/s', 'frac': executed_tflops / PEAK_FP32_MFMA_TFLOPS,
            'direct_equivalent_tflops': direct_tflops,
            'direct_equivalent_x_peak': direct_tflops / PEAK_FP32_MFMA_TFLOPS}
    if abs(direct_tflops - executed_tflops) > 1e-9 * max(direct_tflops, 1.0):
        f43 = 'wino4' in symbol
        roof['algorithm'] = (('fused Winograd F(4x4,3x3): 36 multiplies per 4x4 outputs and (ci,co) instead of 144; '
                              if f43 else
                              'fused Winograd F(2x2,3x3): 16 multiplies per 2x2 outputs and (ci,co) instead of 36; ')
                             + 'achieved = executed flops, direct_equivalent = SURVEY 8(d) algorithmic flops')
    traffic, src = measured_traffic(symbol)
    roof['traffic'] = traffic
    roof['traffic_source'] = None if traffic is None else \
        src + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, committed; not re-measured in this run)'
    return roof


def live_traffic(symbol, args):
    """HBM bytes per launch of `symbol`, measured now: this script re-run as a short child (three forward passes of
    the same program, `--traffic-child`) under rocprofv3 with ONE counter per run (FETCH_SIZE and WRITE_SIZE do not
    fit one pass on gfx950), counters in their own runs with --kernel-trace only, FETCH_SIZE doubled
    (MI355X_MICROARCH.md, HBM section).  Returns (bytes, description) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pmc_traffic
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    vals = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = tempfile.mkdtemp(prefix='egn_pmc_', dir='/tmp')
        cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable,
               os.path.abspath(__file__), '--traffic-child', '--batch', str(args.batch), '--head', args.head]
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=120)
            if r.returncode != 0:
                return None, 'rocprofv3 --pmc %s exited with %d' % (counter, r.returncode)
            per = pmc_traffic.per_kernel(out, counter)
            if symbol not in per:
                return None, '%s not in the %s pass' % (symbol, counter)
            vals[counter] = sum(per[symbol]) / len(per[symbol])
        except subprocess.TimeoutExpired:
            return None, 'rocprofv3 --pmc %s timed out' % counter
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0, \
        'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate child passes of the ' \
        'same forward program, FETCH_SIZE doubled for gfx950)'


def traffic_child(args):
    """What live_traffic() profiles: the W48 forward + decode program of the bench, three passes."""
    from egonet_amd import synth
    torch.cuda.set_device(0)
    cfg, ego, _, _ = build_model(args.head, torch.device('cuda', 0))
    crops = synth.synth_crops(args.batch, 3, 256, 256, seed=100).cuda()
    eng = ego.HC._hip_engine()
    for _ in range(3):
        eng.forward(crops, decode_mode=1 if args.head == 'heatmap' else None)
    torch.cuda.synchronize()


def cpu_baseline(cfg, hc_sd, l_sd, stats, head, seconds):
    """The oracle on this host's cores, bounded sample of the same workload."""
    from egonet_amd import synth
    from oracle import hrnet_oracle, decode_oracle, lifter_oracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1

    def one(x):
        b = len(x)
        out = hrnet_oracle.hrnet_forward(hc_sd, cfg, x)
        maps = (out[0] if isinstance(out, tuple) else out).numpy()
        xy, _ = decode_oracle.soft_arg_max(maps)
        kp = (xy * 4.0).reshape(b, -1).astype(np.float64) + 300.0
        lifter_oracle.lift_2d_to_3d(l_sd, stats, kp)

    # oneDNN convolutions on ~300 small layers do not scale to hundreds of
    # threads (all 256 host threads measured 0.03 crops/s): probe a few thread
    # counts on one crop (bounded) and keep the fastest.
    probe = synth.synth_crops(1, 3, 256, 256, seed=3)
    best_t, best_dt = 1, None
    for t in [t for t in (8, 16, 32, 64) if t <= avail] or [avail]:
        torch.set_num_threads(t)
        one(probe)                           # warm-up (primitive creation)
        t0 = time.time()
        one(probe)
        dt = time.time() - t0
        if best_dt is None or dt < best_dt:
            best_t, best_dt = t, dt
        if dt > 5.0:
            break
    torch.set_num_threads(best_t)
    b = 8
    x = synth.synth_crops(b, 3, 256, 256, seed=3)
    one(x)
    t0 = time.time()
    reps = 0
    while True:
        one(x)
        reps += 1
        if time.time() - t0 >= seconds or reps >= 50:
            break
    dt = time.time() - t0
    return {'value': b * reps / dt, 'unit': 'crops/s', 'cores': best_t, 'kind': 'port',
            'sample': '%d x batch %d crops, HRNet-W48 %s head + soft-arg-max + lifter, CPU oracle (torch '
                      'fp32, %d of %d host threads -- fastest of a 8/16/32/64 probe), %.1f s'
                      % (reps, b, head, best_t, avail, dt)}


GFLOP_FWD_PER_CROP = 42.035      # HRNet-W48 coordinates head, forward, 2*MAC


def _timed(step, steps, warmup, dev, dist):
    """warmup untimed + `steps` timed calls, barrier + synchronize on both sides, max over ranks.
    Like the product's hot loop (egonet_amd.trainer.train freezes after its first iteration) the objects
    alive after the warm-up go to the garbage collector's permanent generation: a full collection costs
    35 ms of host time otherwise and lands in one step out of ~10."""
    import gc
    for _ in range(warmup):
        step()
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    gc.unfreeze()
    return dt / steps, out


def _dominant(timing):
    """`roofline` of the kernel symbol with the largest total time among the hipEvent-bracketed conv / GEMM
    launches of one step (launches, avg_us, share of the timed launches beside it)."""
    torch.cuda.synchronize()
    by = {}
    stats_sym = {'void conv_wino4_kernel<0>(ConvArgs)': 'void conv_wino4s_kernel<0, 1>(ConvArgs)',
                 'void conv_wino4b_kernel<0>(ConvArgs)': 'void conv_wino4s_kernel<1, 1>(ConvArgs)',
                 'void conv_wino4bk_kernel<0>(ConvArgs)': 'void conv_wino4s_kernel<1, 2>(ConvArgs)',
                 'void conv_wino4c_kernel<0, 1>(ConvArgs)': 'void conv_wino4s_kernel<2, 1>(ConvArgs)',
                 'void conv_wino4c_kernel<0, 2>(ConvArgs)': 'void conv_wino4s_kernel<2, 2>(ConvArgs)'}
    for rec in timing:
        cfg, flops, e0, e1 = rec[:4]
        sym = _symbol(cfg) or 'cfg%d' % cfg
        if len(rec) > 4 and rec[4]:          # the training tape's build with BatchNorm statistics in the item end
            sym = stats_sym.get(sym, sym)
        a = by.setdefault(sym, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += flops
        a[3] += flops * _executed(cfg)
    total = sum(a[1] for a in by.values())
    name, (n, ms, fl, xfl) = max(by.items(), key=lambda kv: kv[1][1])
    roof = mfma_roofline(name, fl / (ms * 1e-3) / 1e12, xfl / (ms * 1e-3) / 1e12)
    roof.update(launches=n, avg_us=ms * 1e3 / n, share_of_conv_time=ms / total if total else 0.0,
                conv_ms_per_step=total, algorithmic_gflop_per_launch=fl / n / 1e9)
    return roof


def train_hc_block(args, world, rank, dev, dist):
    """BASELINE configs[3]: train_IGRs HRNet-W48 fwd + bwd + Adam (libs/trainer/trainer.py:183-209)."""
    from egonet_amd import configs, parallel, synth
    from egonet_amd.model.heatmapModel import hrnet
    from egonet_amd.train_hrnet import HRNetTrainStep
    B = args.train_batch
    cfg = configs.w48_config('coordinates')
    net = hrnet.get_pose_net(cfg, is_train=False)
    sd = synth.synth_state_dict(net.state_dict(), seed=1)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    if dist:
        parallel.broadcast_module(net, src=0)
    tr = HRNetTrainStep(net, lr=1e-3, grad_sync=parallel.FlatGradSync(32.0) if dist else None)
    g = torch.Generator().manual_seed(100 + rank)
    x = synth.synth_crops(B, 3, 256, 256, seed=50 + rank).to(dev)
    tgt = torch.rand(B, 33, 64, 64, generator=g).to(dev)
    jt = (torch.rand(B, 33, 2, generator=g) * 256).to(dev)
    sec, loss = _timed(lambda: tr.step(x, tgt, jt), args.steps, max(args.warmup, 2), dev, dist)
    out = None
    if rank == 0:
        # one extra step on ONE stream with hipEvents around every forward / data-gradient conv --
        # rank 0 alone, so WITHOUT the gradient exchange (a collective only one rank enters would hang)
        side, tr.wgrad_stream = tr.wgrad_stream, None
        sync, tr.grad_sync = tr.grad_sync, None
        tr.timing = []
        tr.step(x, tgt, jt, update=False)
        roof = _dominant(tr.timing)
        tr.timing, tr.wgrad_stream, tr.grad_sync = None, side, sync
        crops = B * world
        out = {'metric': 'hc_train_crops_per_sec', 'value': crops / sec, 'unit': 'crops/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': max(args.warmup, 2), 'ms_per_step': sec * 1e3, 'scaling': 'weak',
               'dtype': 'f32', 'data': 'synthetic', 'loss': float(loss.item()),
               'direct_equivalent_tflops_per_gpu': 3 * GFLOP_FWD_PER_CROP * B / sec / 1e3,
               'roofline': roof,
               'config': {'workload': 'configs[3]: train_IGRs HRNet-W48 256x256 fwd+bwd+Adam, composite loss '
                                      '(mse + 0.1 l1), batch=%d crops/GPU' % B, 'global_batch': crops,
                          'parallelism': 'dp%d, one flat-gradient all-reduce per step (RCCL)' % world
                          if world > 1 else 'dp1'}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle.hrnet_train_oracle import HRNetTrainOracle
            nb = 2
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            orc = HRNetTrainOracle(sd, cfg, lr=1e-3)
            xc, tc, jc = x[:nb].cpu(), tgt[:nb].cpu(), jt[:nb].cpu()
            orc.step(xc, tc, jc)                                   # warm-up (primitive creation)
            t0, reps = time.time(), 0
            while True:
                orc.step(xc, tc, jc)
                reps += 1
                if time.time() - t0 >= min(args.cpu_seconds, 10.0) or reps >= 40:
                    break
            dt = time.time() - t0
            out['cpu_baseline'] = {'value': nb * reps / dt, 'unit': 'crops/s', 'cores': torch.get_num_threads(),
                                   'kind': 'port', 'sample': '%d iterations of a %d-crop batch (torch autograd fp32 '
                                   'training oracle: forward, composite loss, backward, Adam), %.1f s' % (reps, nb, dt)}
    del tr, net
    torch.cuda.empty_cache()
    return out


def train_lifter_block(args, dev):
    """BASELINE configs[2]: train_lifting FCModel fwd + bwd + Adam, batch 4096, one GPU."""
    from egonet_amd import configs, synth
    from egonet_amd.model import FCmodel
    from egonet_amd.train_lifter import LifterTrainStep
    B = args.lifter_batch
    cfg = configs.w48_config()
    net = FCmodel.get_fc_model(1, cfg, 66, 96)               # dropout 0.5 as shipped
    sd = synth.synth_state_dict(net.state_dict(), seed=2)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(B, 66, generator=g), torch.randn(B, 96, generator=g)
    net = net.to(dev).train()
    tr = LifterTrainStep(net, lr=1e-3)
    xd, yd = x.to(dev), y.to(dev)
    steps = max(args.steps, 50)
    sec, loss = _timed(lambda: tr.step(xd, yd), steps, max(args.warmup, 5), dev, None)
    side, tr.wgrad_stream = tr.wgrad_stream, None
    tr.timing = []
    tr.step(xd, yd)
    roof = _dominant(tr.timing)
    tr.timing, tr.wgrad_stream = None, side
    lin = [m for m in net.modules() if isinstance(m, torch.nn.Linear)]
    fl = sum(2.0 * B * m.in_features * m.out_features * (2 if i == 0 else 3) for i, m in enumerate(lin))
    out = {'metric': 'lifter_train_sets_per_sec', 'value': B / sec, 'unit': 'sets/s', 'n_gpus': 1, 'steps': steps,
           'warmup': max(args.warmup, 5), 'ms_per_step': sec * 1e3, 'dtype': 'f32', 'data': 'synthetic',
           'loss': float(loss.item()), 'gemm_gflop_per_step': fl / 1e9, 'algorithmic_tflops': fl / sec / 1e12,
           'frac_of_fp32_peak': fl / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 'roofline': roof,
           'config': {'workload': 'configs[2]: train_lifting FCModel(66->96, 1024 x 2 blocks) fwd+bwd+Adam, '
                                  'batch=%d, dropout 0.5' % B}}
    if not args.no_cpu_baseline:
        from oracle.lifter_train_oracle import LifterTrainOracle
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        orc = LifterTrainOracle(sd, lr=1e-3)
        orc.step(x, y)
        t0 = time.time()
        for _ in range(3):
            orc.step(x, y)
        dt = (time.time() - t0) / 3
        out['cpu_baseline'] = {'value': B / dt, 'unit': 'sets/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                               'sample': '3 iterations of one %d-set batch (torch autograd fp32 training oracle, '
                                         'dropout off)' % B}
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: re-execute this script as N ranks, one per
    device, under torch.distributed.run (backend nccl = RCCL; the reference's tools/train_IGRs.py:59,111 uses
    every visible GPU from one process instead).  Returns the launcher's exit code."""
    import subprocess
    backend = os.environ.get('EGONET_AMD_DIST_BACKEND', 'nccl')
    if backend == 'nccl' and not args.dry_run:
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) are visible (set EGONET_AMD_DIST_BACKEND=gloo for '
                             'a control-flow smoke test with every rank on device 0)\n' % (args.gpus, have))
            return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC (RCCL across processes on this driver)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """No GPU work: process group, barrier, max-over-ranks reduction of a host clock, the JSON line."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('EGONET_AMD_DIST_BACKEND', 'gloo'))
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps)
    dt = time.perf_counter() - t0
    ranks = 1
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ranks = dist.get_world_size()
    if rank == 0:
        print(json.dumps({'metric': 'crops/sec (256x256, heatmap+decode+lift)', 'value': 0.0, 'unit': 'crops/s',
                          'n_gpus': world, 'ranks': ranks, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': dt / args.steps * 1e3, 'dry_run': True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d\n'
                         % (args.gpus, world, world))
    if args.dry_run:
        return dry_run(args, world, rank)
    if args.traffic_child:
        return traffic_child(args)
    dist = None
    backend = 'none'
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # EGONET_AMD_DIST_BACKEND=gloo: control-flow smoke test of the multi-rank path on a box with fewer
        # GPUs than ranks (every rank on device 0, collectives through the host) -- never a measurement
        backend = os.environ.get('EGONET_AMD_DIST_BACKEND', 'nccl')
        if backend != 'nccl':
            local = 0
        elif torch.cuda.device_count() <= local:
            raise RuntimeError('rank %d: LOCAL_RANK %d but only %d GPU(s) visible' % (rank, local,
                                                                                     torch.cuda.device_count()))
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local if world > 1 else 0)

    from egonet_amd import synth
    from egonet_amd.parallel import shard_range
    cfg, ego, hc_sd, l_sd = build_model(args.head, dev)
    B = args.batch
    # weak scaling: the global batch is world*B crops; this rank owns its contiguous shard
    lo, hi = shard_range(world * B, world, rank)
    crops = synth.synth_crops(B, 3, 256, 256, seed=100 + rank).to(dev)
    boxes = synth.synth_boxes(world * B, seed=5)[lo:hi]
    from egonet_amd.common.img_proc import modify_bbox
    rets = [modify_bbox(b, 1.0) for b in boxes]
    centers = torch.tensor(np.stack([r['c'] for r in rets]), dtype=torch.float64, device=dev)
    scales = torch.tensor(np.stack([r['s'] for r in rets]), dtype=torch.float64, device=dev)
    K = np.array([[707.0493, 0., 604.0814], [0., 707.0493, 180.5066], [0., 0., 1.]])
    decode = 'soft' if args.head == 'heatmap' else 'coords'

    def step():
        return ego.infer_crops(crops, centers, scales, K=K, decode=decode, to_host=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out['kpts_3d']).all()

    # ---- the same K steps with TWO batches in flight (serving form): two streams alternate, each with its own copy of
    # the launch programs (slot 0 / 1: own arena, own launch lanes), so that the low-occupancy head and tail of one batch
    # (stem / final 1x1 / decode / lifter / pose solve) and every kernel's last wave overlap the other batch's kernels.
    # Reported BESIDE the headline (`value` stays one batch at a time, comparable with the earlier rounds).
    two = None
    if world == 1 and args.pipelined:
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        for st_ in streams:
            st_.wait_stream(torch.cuda.current_stream(dev))

        def pstep(i):
            with torch.cuda.stream(streams[i & 1]):
                return ego.infer_crops(crops, centers, scales, K=K, decode=decode, to_host=False, slot=i & 1)
        for i in range(max(args.warmup, 4)):
            last = pstep(i)
        torch.cuda.synchronize()
        assert torch.equal(last['kpts_3d'], out['kpts_3d'])          # the same bits as the one-stream step
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = pstep(i)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        two = {'value': args.steps * B / dt2, 'unit': 'crops/s', 'ms_per_step': dt2 / args.steps * 1e3,
               'in_flight': 2, 'note': 'two streams alternate, a program copy per stream; same K steps, same bits'}

    if rank == 0:
        # per-kernel timing of the backbone program: hipEvents around every launch,
        # serial on the stream the kernels run on, outside the timed region
        eng = ego.HC._hip_engine()
        mode = 1 if decode == 'soft' else None
        samples = []
        for it in range(5):
            eng.forward(crops, decode_mode=mode, timed=True)
            if it >= 2:
                samples.append(eng.last_ms)
        ms = np.mean(samples, axis=0)
        prog = eng.program(crops, mode)
        (rows, total_ms), (syms, _) = kernel_tables(prog, ms)
        dom = syms[0]
        mfma_bound = dom['flops'] > 0 and dom['flops'] / max(dom['bytes'], 1) > 20
        if mfma_bound:
            roof = mfma_roofline(dom['name'], dom['tflops'], dom['executed_tflops'])
        else:
            traffic, src = measured_traffic(dom['name'])
            roof = {'bound': 'hbm', 'kernel': dom['name'], 'achieved': dom['gbps'], 'peak': PEAK_HBM_GBPS,
                    'unit': 'GB/s', 'frac': dom['gbps'] / PEAK_HBM_GBPS, 'traffic': traffic,
                    'traffic_source': None if traffic is None else src + ' (committed rocprofv3 --pmc passes)'}
        roof.update(launches=dom['launches'], avg_us=dom['avg_us'], time_share=dom['share'],
                    algorithmic_gflop_per_launch=dom['flops'] / dom['launches'] / 1e9,
                    algorithmic_mb_per_launch=dom['bytes'] / dom['launches'] / 1e6)
        if args.live_traffic and world == 1:
            lt, how = live_traffic(dom['name'], args)
            if lt is not None:
                roof['traffic'], roof['traffic_source'] = lt, how
            else:
                roof['traffic_live_error'] = how
        conv_flops = sum(a['flops'] for a in rows)
        conv_xflops = sum(a['xflops'] for a in rows)

        def slim(a):
            return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in a.items()
                    if k in ('name', 'launches', 'ms', 'share', 'tflops', 'executed_tflops', 'gbps', 'avg_us')}
        value = world * B * args.steps / dt
        result = {
            'metric': 'crops/sec (256x256, heatmap+decode+lift)', 'value': value, 'unit': 'crops/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: batch=%d 256x256 crops/GPU, HRNet-W48 %s head forward + '
                                   '%s decode + affine + FC lifter + pose solve' % (B, args.head, decode),
                       'global_batch': world * B, 'weights': 'synthetic (egonet_amd.synth, seeded per key)',
                       'parallelism': 'replicas x%d, crops sharded by rank, no collective' % world,
                       'ranks': dist.get_world_size() if dist else 1, 'backend': backend,
                       'launch_lanes': int(eng.lanes)},
            'roofline': roof,
            'backbone': {'ms_sum_of_kernels': total_ms, 'gflop_per_crop': conv_flops / B / 1e9,
                         'direct_equivalent_tflops_overall': conv_flops / (total_ms * 1e-3) / 1e12,
                         'executed_tflops_overall': conv_xflops / (total_ms * 1e-3) / 1e12,
                         'executed_frac_of_fp32_peak': conv_xflops / (total_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         'launches': int(sum(a['launches'] for a in rows)), 'arena_mb': prog.arena_bytes / 2 ** 20,
                         'weights_mb': prog.weight_bytes / 2 ** 20},
            'kernel_symbols': [slim(a) for a in syms[:8]],
            'kernels': [slim(a) for a in rows[:12]],
        }
        if two is not None:
            result['two_batches_in_flight'] = two
        if args.profile_json:
            with open(args.profile_json, 'w') as f:
                json.dump({'ops': [dict(m, ms=float(t)) for m, t in zip(prog.meta, ms)], 'classes': rows,
                           'symbols': syms}, f, indent=1)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(cfg, hc_sd, l_sd, ego.LS, args.head, args.cpu_seconds)
    # ---- the training configurations, same run, their own timed regions ----
    if not args.no_train:
        del ego, crops
        torch.cuda.empty_cache()
        try:
            hc_block = train_hc_block(args, world, rank, dev, dist)
        except Exception as e:          # never lose the headline line to the extra blocks
            hc_block = {'error': '%s: %s' % (type(e).__name__, e)}
        if rank == 0:
            result['train_hc'] = hc_block
            if world == 1:
                try:
                    result['train_lifter'] = train_lifter_block(args, dev)
                except Exception as e:
                    result['train_lifter'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
