"""Measured choice of the conv tile configuration ("measure, don't guess").

For every distinct convolution shape of a program the engine asks ``choose``:
  1. a shipped table (``tuned/gfx950.json``, measured on MI355X, committed) and
     the in-process cache are consulted;
  2. on a miss -- unless ``EGONET_AMD_AUTOTUNE=0`` -- every compiled tile
     configuration that the host planner accepts is timed on the real shape
     (scratch tensors, hipEvents on the current stream, min of 5 after 2
     warm-ups) and the fastest is kept;
  3. with autotuning off the library's cost-model planner decides (cfg 0).
``EGONET_AMD_TUNE_DUMP=<path>`` writes everything tuned in this process as
JSON at exit (that is how the shipped table is produced).
"""
import atexit
import ctypes as C
import json
import os

import torch

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
TABLE_PATH = os.path.join(_HERE, 'tuned', 'gfx950.json')
_table = None
_tuned_here = {}


def _load():
    global _table
    if _table is None:
        _table = {}
        if os.path.isfile(TABLE_PATH) and os.environ.get('EGONET_AMD_RETUNE', '0') != '1':
            try:
                with open(TABLE_PATH) as f:
                    _table = {k: v for k, v in json.load(f).items() if not k.startswith('_')}
            except (OSError, ValueError):
                _table = {}
    return _table


def shape_key(n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride, pad, has_res, out_nchw):
    return 'n%d_h%d_w%d_ci%d.%d_co%d.%d_k%dx%d_s%d_p%d_r%d_o%d' % (
        n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride, pad, int(has_res), int(out_nchw))


def autotune_enabled():
    return os.environ.get('EGONET_AMD_AUTOTUNE', '1') != '0'


def _time_cfg(L, args, cfg, stream, x, w, sc, sh, res, y):
    """One conv of shape ``args`` under configuration ``cfg`` as a ONE-OP PROGRAM -- the way the engine launches it
    (configurations that split a layer over several kernels when called through egn_conv2d_f32 run as one launch inside
    a program: conv_wino4c_kernel's ticket words belong to the program's op).  Min of 5 after 2 warm-ups, ms."""
    n, h, wd, cin, cs_in, cout, cs_out, kh, kw, stride, pad, has_res, out_nchw = args
    prog = C.c_void_p(L.egn_program_create(8))
    if not prog:
        return None
    try:
        refs = []
        for slot, t in enumerate((x, w, sc, sh, res if has_res else None, y)):
            if t is None:
                refs.append(_lib.NULL_REF)
                continue
            if L.egn_program_bind(prog, slot, _lib.ptr(t)) != 0:
                return None
            refs.append(_lib.Ref(slot, 0))
        if L.egn_program_add_conv2d(prog, *refs, n, h, wd, cin, cs_in, cout, cs_out, kh, kw, stride, pad, 1,
                                    int(out_nchw), cfg) != 0:
            return None

        def launch():
            return L.egn_program_run(prog, stream)
        if launch() != 0:
            return None
        launch()
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None or t < best else best
        return best
    finally:
        torch.cuda.synchronize()
        L.egn_program_destroy(prog)


def tune(device, args, skip=(), only=None):
    """Time every tile configuration on the real shape (except ``skip``; ``only``: just these ids); returns
    (cfg, {cfg: ms})."""
    L = _lib.lib()
    n, h, wd, cin, cs_in, cout, cs_out, kh, kw, stride, pad, has_res, out_nchw = args
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    coutp = (cout + 15) // 16 * 16
    nchunk = (cin + 15) // 16
    g = torch.Generator(device='cpu').manual_seed(1)
    with torch.cuda.device(device):
        x = torch.randn(n * h * wd * cs_in, device=device)
        # one buffer serves every packing (random data: only the timing matters); the Winograd kernels read 16
        # (F(2x2,3x3)) / 36 or 48 (F(4x4,3x3), conv_wino4_kernel's padded layout) floats per (co, ci)
        w = torch.randn(max(nchunk * kh * kw * 4 * coutp * 4, cout * cin * 48), device=device) * 0.05
        sc = torch.ones(coutp, device=device)
        sh = torch.zeros(coutp, device=device)
        ny = n * ho * wo * (cout if out_nchw else cs_out)
        y = torch.empty(ny, device=device)
        res = torch.randn(ny, device=device) if has_res else None
        stream = _lib.current_stream(device)
        times = {}
        for cfg in range(1, L.egn_conv_num_configs() + 1):
            if L.egn_conv_config_kind(cfg) < 0 or cfg in skip or (only is not None and cfg not in only):
                continue                                              # (kind < 0: timing-ablation builds)
            out = (C.c_int * 12)()
            if L.egn_conv_plan_query(n, h, wd, cin, cs_in, cout, cs_out, kh, kw, stride, pad,
                                     int(out_nchw), cfg, out) != 0:
                continue
            t = _time_cfg(L, args, cfg, stream, x, w, sc, sh, res, y)
            if t is not None:
                times[cfg] = t
        torch.cuda.synchronize(device)
    del g
    if not times:
        return 0, {}
    return min(times, key=times.get), times


def _pick(entry, allow_wino, allow_f43=False):
    """Fastest measured configuration of a table entry among the kernel KINDS the caller can feed
    (egn_conv_config_kind: 0 direct-packed filter -- always; 1 Winograd F(2x2,3x3) with ``allow_wino``;
    2 Winograd F(4x4,3x3) with ``allow_f43``: the inference engine, which transforms filters on the host)."""
    L = _lib.lib()
    ok = {0} | ({1} if allow_wino else set()) | ({2, 3} if allow_f43 else set())
    # EGONET_AMD_SKIP_CFG=86,84: same-box A/B runs of a new configuration against the table without it
    skip = {int(v) for v in os.environ.get('EGONET_AMD_SKIP_CFG', '').split(',') if v.strip()}
    cfg = int(entry['cfg'])
    if cfg <= 0 or (L.egn_conv_config_kind(cfg) in ok and cfg not in skip):
        return cfg
    fit = {int(k): v for k, v in entry.get('ms', {}).items()
           if L.egn_conv_config_kind(int(k)) in ok and int(k) not in skip}
    return min(fit, key=fit.get) if fit else 0


def _forced_wino(args, kind=1, order=None):
    """EGONET_AMD_WINO=1 / =43 / =43b: the Winograd F(2x2,3x3) / F(4x4,3x3) configuration for every shape one plans
    for (parity tests pin the kernel families on the same fixtures); returns 0 if none does.  ``order``: the ids to
    try first (=43 prefers conv_wino4_kernel's 16 x 32 regions, cfg 70, and takes conv_wino4b_kernel, cfg 80, for the
    maps only it plans; =43b prefers cfg 80 everywhere)."""
    n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride, pad, has_res, out_nchw = args
    L = _lib.lib()
    out = (C.c_int * 12)()
    for cfg in list(order or ()) + list(range(L.egn_conv_num_configs(), 0, -1)):
        if L.egn_conv_config_kind(cfg) == kind and L.egn_conv_plan_query(
                n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride, pad, int(out_nchw), cfg, out) == 0:
            return cfg
    return 0


def choose(device, args, allow_wino=False, allow_f43=False):
    """``allow_wino`` / ``allow_f43``: the caller packs the filter for whatever configuration kind comes back
    (egn_conv_config_kind 1 / 2).  EGONET_AMD_WINO=0 never returns a Winograd configuration, =1 always the
    F(2x2,3x3) one where it plans, =43 / =43b an F(4x4,3x3) one where one plans (cfg 70 / cfg 80 first; else
    F(2x2,3x3)); EGONET_AMD_F43=0 keeps F(4x4,3x3) out; default: whichever measured fastest."""
    mode = os.environ.get('EGONET_AMD_WINO', '')
    if os.environ.get('EGONET_AMD_F43', '1') == '0' or mode in ('0', '1'):
        allow_f43 = False
    if mode == '0':
        allow_wino = False
    elif mode in ('43', '43b') and allow_f43:
        cfg = _forced_wino(args, 3, (70, 80) if mode == '43' else (80, 70)) or _forced_wino(args, 2) or \
            (_forced_wino(args, 1) if allow_wino else 0)
        if cfg:
            return cfg
    elif mode in ('1', '43', '43b') and allow_wino:
        cfg = _forced_wino(args)
        if cfg:
            return cfg
    key = shape_key(*args)
    tab = _load()
    only = os.environ.get('EGONET_AMD_F43_MATCH', '')        # debugging: F(4x4,3x3) only for shape keys containing this
    if only and only not in key:
        allow_f43 = False
    if key in tab:
        return _pick(tab[key], allow_wino, allow_f43)
    if not autotune_enabled():
        return 0               # deterministic: the shipped table or the cost model, whatever was tuned earlier
    if key in _tuned_here:
        return _pick(_tuned_here[key], allow_wino, allow_f43)
    cfg, times = tune(device, args)
    _tuned_here[key] = {'cfg': cfg, 'ms': {str(k): round(v, 5) for k, v in times.items()}}
    return _pick(_tuned_here[key], allow_wino, allow_f43)


def tuned_in_process():
    return dict(_tuned_here)


def _dump():
    path = os.environ.get('EGONET_AMD_TUNE_DUMP')
    if path and _tuned_here:
        merged = dict(_load())
        merged.update(_tuned_here)
        try:
            with open(path, 'w') as f:
                json.dump(merged, f, indent=0, sort_keys=True)
        except OSError:
            pass


atexit.register(_dump)
