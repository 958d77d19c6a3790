"""Kernel-level parity on a real MI355X: every HIP kernel, called through the C
ABI, against the CPU oracle (torch fp32 / numpy restatements) on seeded inputs.

Tolerances: fp32 conv / GEMM outputs 2e-4 abs on O(1) activations (the budget
of BASELINE.json is 1e-3 on key-points), arg-max indices bit exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import decode_oracle, geometry_oracle

pytestmark = pytest.mark.gpu


def _bn(cout, g):
    bn = torch.nn.BatchNorm2d(cout)
    bn.weight.data = 0.5 + torch.rand(cout, generator=g)
    bn.bias.data = torch.randn(cout, generator=g) * 0.2
    bn.running_mean = torch.randn(cout, generator=g) * 0.2
    bn.running_var = 0.5 + torch.rand(cout, generator=g)
    return bn.eval()


def _conv_case(n, h, w, cin, cout, k, s, p, act=1, use_res=False, nchw=False, cfg=0, seed=0, bias=False):
    from egonet_amd import ops
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / np.sqrt(cin * k * k) * 1.7
    b = torch.randn(cout, generator=g) * 0.3 if bias else None
    bn = _bn(cout, g)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    res = torch.randn(n, cout, ho, wo, generator=g) if use_res else None
    with torch.no_grad():
        ref = bn(F.conv2d(x, wt, b, s, p))
        a = act & 0xf
        f = {0: lambda t: t, 1: F.relu, 2: torch.sigmoid, 3: lambda t: F.leaky_relu(t, 0.01)}[a]
        if res is not None and not (act & 0x10):
            ref = ref + res
        ref = f(ref)
        if res is not None and (act & 0x10):
            ref = res + ref
    from egonet_amd import _lib
    kind = _lib.lib().egn_conv_config_kind(cfg) if cfg > 0 else 0
    pc = ops.PackedConv(wt, b, bn, wino=(kind == 1), kind=kind if kind in (2, 3) else None)
    xd = ops.nchw_to_nhwc(x.cuda())
    rd = ops.nchw_to_nhwc(res.cuda()) if res is not None else None
    y = ops.conv2d_nhwc(xd, pc, cin, s, p, act, rd, out_nchw=nchw, cfg=cfg)
    torch.cuda.synchronize()
    if nchw:
        got = y.cpu()
    else:
        assert float(y[..., cout:].abs().max() if y.shape[-1] > cout else 0.0) == 0.0
        got = ops.nhwc_to_nchw(y, cout).cpu()
    err = (got - ref).abs().max().item()
    assert torch.isfinite(got).all()
    return err


# the shape classes of HRNet-W48 @256x256 at small batch + edge shapes
HRNET_SHAPES = [
    # n   h    w   cin  cout k  s  p
    (2, 64, 64, 48, 48, 3, 1, 1),
    (2, 32, 32, 96, 96, 3, 1, 1),
    (2, 16, 16, 192, 192, 3, 1, 1),
    (3, 8, 8, 384, 384, 3, 1, 1),
    (2, 64, 64, 64, 64, 3, 1, 1),
    (1, 64, 64, 256, 48, 3, 1, 1),
    (1, 64, 64, 256, 96, 3, 2, 1),
    (2, 256, 256, 3, 64, 3, 2, 1),
    (2, 128, 128, 64, 64, 3, 2, 1),
    (2, 64, 64, 48, 96, 3, 2, 1),
    (2, 16, 16, 192, 384, 3, 2, 1),
    (2, 64, 64, 64, 256, 1, 1, 0),
    (2, 64, 64, 256, 64, 1, 1, 0),
    (2, 32, 32, 96, 48, 1, 1, 0),
    (2, 8, 8, 384, 48, 1, 1, 0),
    (2, 64, 64, 35, 66, 3, 2, 1),
    (2, 64, 64, 35, 66, 1, 2, 0),
    (3, 4, 4, 66, 66, 3, 1, 1),
]


@pytest.mark.parametrize('shape', HRNET_SHAPES)
def test_conv_hrnet_shapes(shape):
    n, h, w, cin, cout, k, s, p = shape
    err = _conv_case(n, h, w, cin, cout, k, s, p, act=1, use_res=(s == 1 and k == 3), seed=cin + cout)
    assert err < 2e-4, err


@pytest.mark.parametrize('cfg', list(range(1, 31)))      # 1..10 staged, 11..30 LDS-DMA (31..40 retired)
def test_conv_every_tile_config(cfg):
    # odd sizes: partial tiles in x, y, batch and channels
    import ctypes as C
    from egonet_amd import _lib
    err = _conv_case(3, 19, 13, 20, 40, 3, 1, 1, act=1, use_res=True, cfg=cfg, seed=cfg)
    assert err < 2e-4, (cfg, err)
    # stride 2: the halo of a 128/256-row tile exceeds the per-lane staging depth
    # (8 dwordx4); the planner must then refuse the forced config cleanly (-2)
    out = (C.c_int * 12)()
    rc = _lib.lib().egn_conv_plan_query(2, 11, 9, 37, 40, 70, 72, 3, 3, 2, 1, 0, cfg, out)
    assert rc in (0, -2)
    if rc == 0:
        err = _conv_case(2, 11, 9, 37, 70, 3, 2, 1, act=0, cfg=cfg, seed=50 + cfg)
        assert err < 2e-4, (cfg, err)
    else:
        assert (cfg - 1) % 10 < 6      # only the >= 128-row tiles overflow
    assert _lib.lib().egn_conv_plan_query(3, 19, 13, 20, 20, 40, 40, 3, 3, 1, 1, 0, 31 + (cfg - 1) % 10, out) != 0   # retired ids never plan


@pytest.mark.parametrize('n,h,w,act', [(2, 256, 256, 1), (3, 64, 48, 1), (1, 34, 22, 0), (5, 32, 32, 3)])
def test_conv_stem_kernel(n, h, w, act):
    """cfg 64 (csrc/conv_stem.hip): the 3 -> 64 channel 3x3 stride-2 stem (hrnet.py:311-314) with K = (tap, channel)
    -- full size, ragged tiles (maps that are not multiples of 16), more tiles than blocks, LeakyReLU."""
    import ctypes as C
    from egonet_amd import _lib
    L = _lib.lib()
    out = (C.c_int * 12)()
    assert L.egn_conv_config_kind(64) == 0
    assert L.egn_conv_plan_query(n, h, w, 3, 4, 64, 64, 3, 3, 2, 1, 0, 64, out) == 0
    err = _conv_case(n, h, w, 3, 64, 3, 2, 1, act=act, cfg=64, seed=h)
    assert err < 2e-4, err
    # what it must refuse: stride 1, other widths, a residual-after activation
    assert L.egn_conv_plan_query(n, h, w, 3, 4, 64, 64, 3, 3, 1, 1, 0, 64, out) != 0
    assert L.egn_conv_plan_query(n, h, w, 3, 4, 48, 48, 3, 3, 2, 1, 0, 64, out) != 0
    assert L.egn_conv_plan_query(n, h, w, 16, 16, 64, 64, 3, 3, 2, 1, 0, 64, out) != 0


@pytest.mark.parametrize('n,h,w,cin,cout,res,act', [
    (2, 64, 64, 48, 48, True, 1),      # the HRNet shape classes at small batch
    (2, 32, 32, 96, 96, True, 1),
    (2, 16, 16, 192, 192, False, 1),
    (5, 8, 8, 384, 384, True, 1),      # 8 x 8 maps: partial image batch for the 4-image tiles
    (1, 16, 16, 16, 48, False, 0),     # one tile, one chunk, every halo side is padding, no activation
    (3, 24, 40, 32, 96, True, 1),      # partial tiles in x and y (24 = 16 + 8, 40 = 32 + 8)
    (70, 32, 32, 48, 48, True, 1),     # more work items than one round of persistent blocks
    (9, 6, 4, 64, 144, False, 1),      # maps smaller than a tile, 3 co-tiles
    (2, 64, 48, 32, 32, True, 1),      # the W32 / Pedestrian widths: 32-channel co-tiles (8-wave kernels only)
    (2, 32, 24, 64, 64, True, 1),
    (3, 16, 12, 128, 128, False, 0),
    (5, 8, 6, 256, 256, True, 1),
    (2, 64, 64, 64, 64, False, 1),     # the 64-channel stem / layer1 3x3 convs of every model
])
def test_conv_winograd_kernels(n, h, w, cin, cout, res, act):
    """Configs 51 / 52 (8 waves, frequency halves + partial exchange), 56 / 57, 59..62 of csrc/conv_wino.hip: fused
    Winograd F(2x2,3x3); the filter transform runs on the device (egn_wino_pack_weight_f32).  Same oracle and tolerance
    as the direct kernels.  45 / 46 (the 4-wave kernel) and 67 / 68 (two 4-wave blocks per CU) were measured and
    retired: they are tested when the library under test is a probe build (-DEGN_PROBES), refused otherwise."""
    import ctypes as C
    from egonet_amd import _lib
    L = _lib.lib()
    probes = bool(L.egn_probe_build())
    assert [L.egn_conv_config_kind(c) for c in (44, 51, 52, 56, 57, 59, 60, 61, 62)] == [0] + [1] * 8
    assert [L.egn_conv_config_kind(c) for c in (45, 46, 67, 68)] == [1 if probes else -1] * 4
    assert L.egn_conv_config_kind(47) == -1 and L.egn_conv_config_kind(53) == -1     # timing ablations: never selectable
    assert L.egn_conv_config_kind(58) == -1 and L.egn_conv_config_kind(63) == -1     # stamp builds neither
    out = (C.c_int * 12)()
    # 56 / 57: 4 waves on 32 tiles (two 8 x 8 images / an 8 x 16 tile); 59..62 = conv_wino9_kernel (round 3: scalar
    # item index math, one instruction stream for both frequency halves) on the geometries of 51 / 52 / 56 / 57;
    # 67 / 68 = conv_wino9_kernel with 8-channel stages (two 4-wave blocks per CU) on the tiles of 62 / 61
    for cfg in (45, 46, 51, 52, 56, 57, 59, 60, 61, 62, 67, 68):
        rc = L.egn_conv_plan_query(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, cfg, out)
        if (cfg in (46, 52, 56, 60, 61, 68) and (h > 8 or w > 8)) or (cfg in (45, 46) and cout % 48) or \
                (cfg in (45, 46, 67, 68) and not probes):
            assert rc != 0
            continue
        assert rc == 0
        err = _conv_case(n, h, w, cin, cout, 3, 1, 1, act=act, use_res=res, cfg=cfg, seed=n + h + cin)
        assert err < 2e-4, (cfg, err)
    # what the planner must refuse: stride 2, 1x1, padded channel strides, Cout not a multiple of 48
    assert L.egn_conv_plan_query(2, 16, 16, 48, 48, 48, 48, 3, 3, 2, 1, 0, 59, out) != 0
    assert L.egn_conv_plan_query(2, 16, 16, 48, 48, 48, 48, 1, 1, 1, 0, 0, 59, out) != 0
    assert L.egn_conv_plan_query(2, 16, 16, 35, 36, 48, 48, 3, 3, 1, 1, 0, 59, out) != 0
    assert L.egn_conv_plan_query(2, 16, 16, 48, 48, 80, 80, 3, 3, 1, 1, 0, 51, out) != 0
    assert L.egn_wino_weight_floats(80, 48, 0) == 0 and L.egn_wino_weight_floats(48, 80, 1) == 0


@pytest.mark.parametrize('n,cin,cout,res,act,nchw', [
    (64, 1024, 1024, True, 1, False),    # the lifter's residual-block layers at the bench batch
    (64, 1024, 1024, False, 1, False),
    (64, 1024, 96, False, 0, True),      # its last layer (hands over [N, C])
    (5, 64, 48, True, 3, False),         # ragged batch, short K, LeakyReLU
    (130, 272, 32, False, 1, False),     # batch over 128, K not a multiple of four chunks
])
def test_conv_fc_kernel(n, cin, cout, res, act, nchw):
    """Config 79, csrc/conv_fc.hip: 1x1 convolution on 1 x 1 maps (Linear + BatchNorm1d + activation + residual of
    libs/model/FCmodel.py:29-52 at inference batch sizes): one 16 x 16 output tile per block, K split over the four
    waves.  Same oracle and tolerance as the general kernels; the planner refuses everything else."""
    import ctypes as C
    from egonet_amd import _lib
    L = _lib.lib()
    out = (C.c_int * 12)()
    assert L.egn_conv_config_kind(79) == 0
    assert L.egn_conv_plan_query(n, 1, 1, cin, cin, cout, cout, 1, 1, 1, 0, int(nchw), 79, out) == 0
    err = _conv_case(n, 1, 1, cin, cout, 1, 1, 0, act=act, use_res=res, nchw=nchw, cfg=79, seed=n + cin)
    assert err < 2e-4, err
    # ... and the 1x1 convs of the fuse layers on the coarse maps: rows = N * H * W
    err = _conv_case(3, 8, 8, 384, 96, 1, 1, 0, act=0, use_res=False, cfg=79, seed=5)
    assert err < 2e-4, err
    assert L.egn_conv_plan_query(4, 2, 2, 64, 64, 48, 48, 1, 1, 1, 0, 1, 79, out) != 0      # NCHW output of a real map
    assert L.egn_conv_plan_query(4, 2, 2, 64, 64, 48, 48, 3, 3, 1, 1, 0, 79, out) != 0      # not 1x1
    assert L.egn_conv_plan_query(4, 1, 1, 66, 68, 48, 48, 1, 1, 1, 0, 0, 79, out) != 0      # Cin % 16
    assert L.egn_conv_plan_query(4, 1, 1, 64, 64, 33, 33, 1, 1, 1, 0, 0, 79, out) != 0      # Cout % 16


@pytest.mark.parametrize('n,h,w,cin,cout,res,act', [
    (2, 16, 32, 16, 48, True, 1),      # one region per image, the shortest K loop (2 stages)
    (3, 32, 64, 48, 96, False, 0),     # 4 regions x 2 co-tiles, no activation, no residual
    (1, 64, 64, 32, 48, True, 1),      # the 64 x 64 maps of stage 2
    (5, 32, 32, 96, 144, True, 1),     # odd batch, 3 co-tiles
])
def test_conv_wino4_kernel(n, h, w, cin, cout, res, act):
    """Config 70, csrc/conv_wino4.hip: fused Winograd F(4x4,3x3) (input transform once per (tile, channel) into
    LDS, filter from global memory into MFMA B registers; host filter transform engine.pack_wino4_weight).
    Same oracle as the direct kernels; F(4x4,3x3) in fp32 carries ~10x the rounding error of F(2x2,3x3)
    (transform constants up to 8, tools/wino43_error_study.py): tolerance 5e-4 on outputs of magnitude ~3."""
    import ctypes as C
    from egonet_amd import _lib
    L = _lib.lib()
    assert L.egn_conv_config_kind(70) == 3 and L.egn_conv_config_kind(71) == -1 and L.egn_conv_config_kind(78) == -1
    assert L.egn_wino4_weight_floats(cout, cin) == (cout // 48) * (cin // 8) * 2 * 12 * 3 * 64 * 4
    out = (C.c_int * 12)()
    assert L.egn_conv_plan_query(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 70, out) == 0
    err = _conv_case(n, h, w, cin, cout, 3, 1, 1, act=act, use_res=res, cfg=70, seed=n + h + cin)
    assert err < 5e-4, err
    # what the planner must refuse: maps that are not whole 16 x 32 regions, Cin % 16, Cout % 48, stride 2, 1x1
    assert L.egn_conv_plan_query(2, 8, 32, 48, 48, 48, 48, 3, 3, 1, 1, 0, 70, out) != 0
    assert L.egn_conv_plan_query(2, 16, 16, 48, 48, 48, 48, 3, 3, 1, 1, 0, 70, out) != 0
    assert L.egn_conv_plan_query(2, 16, 32, 24, 24, 48, 48, 3, 3, 1, 1, 0, 70, out) != 0
    assert L.egn_conv_plan_query(2, 16, 32, 48, 48, 64, 64, 3, 3, 1, 1, 0, 70, out) != 0
    assert L.egn_conv_plan_query(2, 32, 64, 48, 48, 48, 48, 3, 3, 2, 1, 0, 70, out) != 0
    assert L.egn_wino4_weight_floats(64, 48) == 0 and L.egn_wino4_weight_floats(48, 20) == 0
    # the product library refuses to launch an ablation / stamp build through the C ABI
    if not L.egn_probe_build():
        import torch
        t = torch.zeros(1 << 20, device='cuda')
        for bad_cfg in (71, 76, 78, 47, 65):
            assert L.egn_conv2d_f32(_lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, _lib.ptr(t), 1, 16, 32, 16, 16,
                                    48, 48, 3, 3, 1, 1, 1, 0, bad_cfg, _lib.current_stream()) != 0, bad_cfg


@pytest.mark.parametrize('n,h,w,cin,cout,res,act', [
    (2, 16, 16, 16, 48, True, 1),      # one region per image, ONE 16-channel stage (the odd-stage tail alone)
    (3, 16, 16, 192, 192, True, 1),    # the 16 x 16 maps of the 192-channel branch: 12 stages, 4 co-tiles
    (2, 64, 64, 48, 48, True, 1),      # 3 stages (odd): pair loop + tail; 16 regions per image
    (3, 32, 48, 32, 96, False, 0),     # 2 x 3 regions, 2 co-tiles, no activation, no residual
    (70, 16, 16, 48, 96, True, 1),     # more work items than one round of persistent blocks (2 items per block)
    (5, 32, 32, 96, 144, False, 1),    # odd batch, 3 co-tiles, 6 stages
])
def test_conv_wino4b_kernel(n, h, w, cin, cout, res, act):
    """Config 80, conv_wino4b_kernel: the F(4x4,3x3) body of csrc/conv_wino4.hip on 16 x 16 pixel regions with
    16-channel stages (one m-tile, k-group pairs) -- the geometry that gives the 16 x 16 maps (hrnet.py:68-92 at
    192 channels) and small batches enough work items.  Same filter pack, oracle and tolerance as config 70."""
    import ctypes as C
    from egonet_amd import _lib
    L = _lib.lib()
    assert L.egn_conv_config_kind(80) == 3 and L.egn_conv_config_kind(81) == -1
    out = (C.c_int * 12)()
    assert L.egn_conv_plan_query(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 80, out) == 0
    assert list(out)[5:8] == [16, 16, 1] and out[10] == n * (h // 16) * (w // 16)
    err = _conv_case(n, h, w, cin, cout, 3, 1, 1, act=act, use_res=res, cfg=80, seed=n + h + cin)
    assert err < 5e-4, err
    # agreement with the 16 x 32 geometry where both plan: the same arithmetic per output -- bit-identical
    if w % 32 == 0:
        import torch
        from egonet_amd import ops
        g = torch.Generator().manual_seed(7)
        x = torch.randn(n, h, w, cin, generator=g).cuda()
        wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        pc = ops.PackedConv(wt, None, None, kind=3)
        ya = ops.conv2d_nhwc(x, pc, cin, 1, 1, 1, None, cfg=70)
        yb = ops.conv2d_nhwc(x, pc, cin, 1, 1, 1, None, cfg=80)
        torch.cuda.synchronize()
        assert torch.equal(ya, yb)
    assert L.egn_conv_plan_query(2, 8, 8, 48, 48, 48, 48, 3, 3, 1, 1, 0, 80, out) != 0
    assert L.egn_conv_plan_query(2, 16, 24, 48, 48, 48, 48, 3, 3, 1, 1, 0, 80, out) != 0


@pytest.mark.parametrize('n,h,w,cin,cout,res,act', [
    (2, 16, 16, 32, 96, True, 1),       # one region per image, 2 stages, one co-tile pair
    (64, 32, 32, 96, 96, True, 1),      # the 96-channel branch at BASELINE's 64 crops: 256 items, 6 stages
    (3, 16, 16, 192, 192, True, 1),     # two co-tile pairs, 12 stages
    (3, 32, 48, 48, 96, False, 0),      # 2 x 3 regions, 3 stages (odd: pair loop + tail), no activation, no residual
    (70, 16, 16, 48, 192, True, 1),     # more items than one round of persistent blocks; pairs on the item axis
    (5, 32, 32, 96, 384, False, 1),     # four co-tile pairs (item mode 1: pairs on the XCD axis), odd batch
    (2, 16, 32, 32, 768, True, 1),      # eight co-tile pairs (item mode 2)
])
def test_conv_wino4w_kernel(n, h, w, cin, cout, res, act):
    """Config 86, conv_wino4w_kernel (csrc/conv_wino4w.hip, round 6): F(4x4,3x3) on 16 x 16 pixel regions with 96 output
    channels (two co-tiles of the same filter pack) per item -- half the input transforms, halo bytes and barriers per
    MFMA of config 80.  The arithmetic per output is config 80's (same transform, same K order, same item end):
    bit-identical to it; oracle and tolerance as config 70 (hrnet.py:68-92 at 96 channels)."""
    import ctypes as C
    from egonet_amd import _lib, ops
    L = _lib.lib()
    assert L.egn_conv_config_kind(86) == 3 and L.egn_conv_config_kind(87) == -1
    out = (C.c_int * 12)()
    assert L.egn_conv_plan_query(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 86, out) == 0
    assert list(out)[5:8] == [16, 16, 1] and out[10] == n * (h // 16) * (w // 16)
    err = _conv_case(n, h, w, cin, cout, 3, 1, 1, act=act, use_res=res, cfg=86, seed=n + h + cin)
    assert err < 5e-4, err
    g = torch.Generator().manual_seed(17)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    pc = ops.PackedConv(wt, None, _bn(cout, g), kind=3)
    r = torch.randn(n, h, w, cout, generator=g).cuda() if res else None
    ya = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=86)
    yb = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=80)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
    # refused: Cout not a multiple of 96, a single 16-channel stage, maps that are not whole 16 x 16 regions, stride 2
    assert L.egn_conv_plan_query(2, 16, 16, 48, 48, 48, 48, 3, 3, 1, 1, 0, 86, out) != 0
    assert L.egn_conv_plan_query(2, 16, 16, 48, 48, 144, 144, 3, 3, 1, 1, 0, 86, out) != 0
    assert L.egn_conv_plan_query(2, 16, 16, 16, 16, 96, 96, 3, 3, 1, 1, 0, 86, out) != 0
    assert L.egn_conv_plan_query(2, 8, 8, 96, 96, 96, 96, 3, 3, 1, 1, 0, 86, out) != 0
    assert L.egn_conv_plan_query(2, 32, 32, 96, 96, 96, 96, 3, 3, 2, 1, 0, 86, out) != 0
    # no training build, no K split: no statistics rows, no ticket words
    assert L.egn_conv2d_bnstats_rows(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 86) == 0
    assert L.egn_conv2d_ticket_words(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 86) == 0


@pytest.mark.parametrize('n,h,w,cin,cout,res,act', [
    (2, 16, 16, 16, 48, True, 1),       # one region per image, two 8-channel stages
    (64, 64, 64, 48, 48, True, 1),      # the 48-channel branch at BASELINE's 64 crops: 1 024 items on 512 blocks
    (3, 16, 16, 192, 192, True, 1),     # four co-tiles (item mode 1), 24 stages
    (3, 32, 48, 32, 96, False, 0),      # 2 x 3 regions, 2 co-tiles, no activation, no residual
    (70, 16, 16, 48, 96, True, 1),      # persistent rounds
    (5, 32, 32, 96, 384, False, 1),     # eight co-tiles (item mode 2), odd batch
])
def test_conv_wino4h_kernel(n, h, w, cin, cout, res, act):
    """Config 88, conv_wino4h_kernel (csrc/conv_wino4h.hip, round 6): F(4x4,3x3) in half-size blocks -- 6 waves, one
    16-tile m-tile x 48 channels, 62 KB of LDS, two independent blocks per CU so that one block's prologue / item end
    runs under the other's MFMAs.  Per output the arithmetic of config 80 (same transform, K order, item end): bit-identical
    to it, with and without the start skew of a CU's second block; oracle and tolerance as config 70 (hrnet.py:49-76)."""
    import ctypes as C
    from egonet_amd import _lib, ops
    L = _lib.lib()
    if not L.egn_probe_build():          # measured slower than config 70 (profiles/r6_wino4h_timeline.txt): probe builds only
        assert L.egn_conv_config_kind(88) == -1 and L.egn_conv_config_kind(90) == -1
        out = (C.c_int * 12)()
        assert L.egn_conv_plan_query(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 88, out) != 0
        pytest.skip('conv_wino4h_kernel / conv_wino4d_kernel are compiled into probe builds only')
    assert L.egn_conv_config_kind(88) == 3 and L.egn_conv_config_kind(89) == -1
    out = (C.c_int * 12)()
    assert L.egn_conv_plan_query(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 88, out) == 0
    assert list(out)[5:8] == [16, 16, 1] and out[10] == n * (h // 16) * (w // 16)
    err = _conv_case(n, h, w, cin, cout, 3, 1, 1, act=act, use_res=res, cfg=88, seed=n + h + cin)
    assert err < 5e-4, err
    g = torch.Generator().manual_seed(19)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    pc = ops.PackedConv(wt, None, _bn(cout, g), kind=3)
    r = torch.randn(n, h, w, cout, generator=g).cuda() if res else None
    ya = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=88)
    yb = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=80)
    # config 90, conv_wino4d_kernel: the two blocks of a CU as the halves of one 12-wave workgroup (LDS-counter barriers)
    assert L.egn_conv_config_kind(90) == 3
    assert L.egn_conv_plan_query(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 90, out) == 0
    yd = [ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=90) for _ in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
    for t in yd:
        assert torch.equal(t, yb)
    assert L.egn_conv_plan_query(2, 16, 16, 24, 24, 48, 48, 3, 3, 1, 1, 0, 88, out) != 0
    assert L.egn_conv_plan_query(2, 8, 8, 48, 48, 48, 48, 3, 3, 1, 1, 0, 88, out) != 0
    assert L.egn_conv_plan_query(2, 32, 32, 48, 48, 48, 48, 3, 3, 2, 1, 0, 88, out) != 0
    assert L.egn_conv2d_bnstats_rows(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 88) == 0
    assert L.egn_conv2d_ticket_words(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 88) == 0


@pytest.mark.parametrize('n,h,w,cin,cout,res,act', [
    (2, 16, 32, 16, 48, True, 1),       # one region per image, two 8-channel stages
    (64, 64, 64, 48, 48, True, 1),      # the 48-channel branch at BASELINE's 64 crops: 512 items
    (3, 16, 32, 192, 192, True, 1),     # four co-tiles (item mode 1), 24 stages
    (3, 32, 64, 32, 96, False, 0),      # 2 x 2 regions, 2 co-tiles, no activation, no residual
    (70, 16, 32, 48, 96, True, 1),      # persistent rounds
    (5, 32, 32, 96, 384, False, 1),     # eight co-tiles (item mode 2), odd batch
    (2, 64, 64, 256, 48, False, 1),     # the 256 -> 48 transition's shape class
])
def test_conv_wino4r_kernel(n, h, w, cin, cout, res, act):
    """Config 92, conv_wino4r_kernel (csrc/conv_wino4r.hip, round 6): F(4x4,3x3) on 16 x 32 regions with row-owner waves --
    the row pass of Y = A^T M A in the accumulators, ONE exchange round per item.  Same transform, K order and filter pack
    as config 70; the two 1-D passes of the output transform run in the other order, so the outputs agree with config 70
    to rounding (not bit for bit); oracle and tolerance as config 70 (hrnet.py:49-76).  Deterministic: two runs, same bits."""
    import ctypes as C
    from egonet_amd import _lib, ops
    L = _lib.lib()
    if not L.egn_probe_build():          # measured slower than config 70 (profiles/r6_wino4r_probe.txt): probe builds only
        assert L.egn_conv_config_kind(92) == -1
        pytest.skip('conv_wino4r_kernel is compiled into probe builds only')
    assert L.egn_conv_config_kind(92) == 3 and L.egn_conv_config_kind(93) == -1
    out = (C.c_int * 12)()
    assert L.egn_conv_plan_query(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 92, out) == 0
    assert list(out)[5:8] == [16, 32, 1] and out[10] == n * (h // 16) * (w // 32)
    err = _conv_case(n, h, w, cin, cout, 3, 1, 1, act=act, use_res=res, cfg=92, seed=n + h + cin)
    assert err < 5e-4, err
    g = torch.Generator().manual_seed(23)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    pc = ops.PackedConv(wt, None, _bn(cout, g), kind=3)
    r = torch.randn(n, h, w, cout, generator=g).cuda() if res else None
    ya = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=92)
    yb = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=92)
    yc = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=70)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
    assert (ya - yc).abs().max().item() < 2e-5 * max(1.0, yc.abs().max().item())
    assert L.egn_conv_plan_query(2, 16, 16, 48, 48, 48, 48, 3, 3, 1, 1, 0, 92, out) != 0      # whole 16 x 32 regions only
    assert L.egn_conv_plan_query(2, 16, 32, 24, 24, 48, 48, 3, 3, 1, 1, 0, 92, out) != 0
    assert L.egn_conv2d_bnstats_rows(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 92) == 0
    assert L.egn_conv2d_ticket_words(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 92) == 0


@pytest.mark.parametrize('n,h,w,cin,cout,res,act', [
    (2, 16, 16, 32, 48, True, 1),       # one stage per half
    (3, 16, 16, 192, 192, True, 1),     # the 16 x 16 maps of the 192-channel branch at a small batch: 6 stages per half
    (16, 16, 16, 192, 192, True, 1),    # BASELINE configs[4]'s per-GPU shard: 64 regions x 4 co-tiles x 2 halves
    (2, 32, 48, 96, 96, False, 0),      # 2 x 3 regions, 3 stages per half (odd), no activation, no residual
    (70, 16, 16, 64, 96, True, 1),      # more item pairs than blocks (persistent rounds), 2 stages per half
])
def test_conv_wino4bk_kernel(n, h, w, cin, cout, res, act):
    """Config 84, conv_wino4bk_kernel: conv_wino4b_kernel (16 x 16 regions) with the input channels of an item split over
    two blocks, the hand-off of conv_wino4c_kernel<., 2> (csrc/conv_wino4.hip) -- small batches of the 96- / 192-channel
    branches have fewer regions than the chip has CUs.  Through egn_conv2d_f32 (zeroed y, atomic adds, finish pass) and as
    a program's op (ticket words, one launch): same bits, twice; against cfg 80 only the order of the two halves' sum."""
    import ctypes as C
    from egonet_amd import _lib, ops
    L = _lib.lib()
    assert L.egn_conv_config_kind(84) == 3
    out = (C.c_int * 12)()
    assert L.egn_conv_plan_query(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 84, out) == 0
    assert list(out)[5:8] == [16, 16, 1]
    err = _conv_case(n, h, w, cin, cout, 3, 1, 1, act=act, use_res=res, cfg=84, seed=n + h + cin)
    assert err < 5e-4, err
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    pc = ops.PackedConv(wt, None, _bn(cout, g), kind=3)
    r = torch.randn(n, h, w, cout, generator=g).cuda() if res else None
    ya = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=84)
    yb = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=84)
    yc = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=80)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
    assert (ya - yc).abs().max().item() < 5e-4
    y = torch.full((n, h, w, cout), float('nan'), device='cuda')
    prog = L.egn_program_create(8)
    assert prog
    try:
        refs = []
        for slot, t in enumerate([x, pc.w, pc.scale, pc.shift, r, y]):
            if t is None:
                refs.append(_lib.NULL_REF)
                continue
            assert L.egn_program_bind(prog, slot, _lib.ptr(t)) == 0
            refs.append(_lib.Ref(slot, 0))
        assert L.egn_program_add_conv2d(prog, *refs, n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, act, 0, 84) == 0
        for _ in range(10):
            y.fill_(float('nan'))
            assert L.egn_program_run(prog, _lib.current_stream()) == 0
            torch.cuda.synchronize()
            assert torch.equal(y, ya)
    finally:
        L.egn_program_destroy(prog)
    assert L.egn_conv_plan_query(2, 16, 16, 48, 48, 48, 48, 3, 3, 1, 1, 0, 84, out) != 0     # whole 16-channel stages per half
    assert L.egn_conv_plan_query(2, 8, 8, 64, 64, 48, 48, 3, 3, 1, 1, 0, 84, out) != 0


@pytest.mark.parametrize('cout,cin', [(48, 16), (96, 48), (192, 192), (384, 384), (48, 256)])
def test_wino4_filter_transform_on_the_device(cout, cin):
    """egn_wino4_pack_weight_f32: U = G g G^T of a torch weight in the register-feed layout of the F(4x4,3x3) kernels,
    computed on the device (float64 arithmetic, one rounding) -- against engine.pack_wino4_weight (host, float64 einsum):
    equal up to the association order (<= 1 ulp of the largest term), padding values exactly zero; dgrad = 1 against the
    host pack of the channel-swapped, tap-rotated weight; and a convolution fed with the device-packed filter against one
    fed with the host pack."""
    from egonet_amd import _lib, engine, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(cout + cin)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    wd = wt.cuda()
    st = _lib.current_stream()
    for dgrad in (0, 1):
        n_out, n_in = (cin, cout) if dgrad else (cout, cin)
        nfl = L.egn_wino4_pack_weight_floats(cout, cin, dgrad)
        if n_out % 48 or n_in % 8:
            assert nfl == 0
            continue
        ref = engine.pack_wino4_weight(wt.permute(1, 0, 2, 3).flip(2, 3).contiguous() if dgrad else wt)
        assert nfl == ref.numel() == L.egn_wino4_weight_floats(n_out, n_in)
        dst = torch.full((nfl,), float('nan'), device='cuda')
        _lib.check(L.egn_wino4_pack_weight_f32(_lib.ptr(wd), cout, cin, dgrad, _lib.ptr(dst), st), 'wino4 pack')
        torch.cuda.synchronize()
        got = dst.cpu()
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() <= 2e-7 * max(1.0, ref.abs().max().item())
        assert torch.equal(got == 0, ref == 0) or (got[ref == 0] == 0).all()
    if cout % 48 == 0 and cin % 16 == 0:
        pc = ops.PackedConv(wt, None, None, kind=3)
        x = torch.randn(2, 32, 32, cin, generator=g).cuda()
        y_host = ops.conv2d_nhwc(x, pc, cin, 1, 1, 0, None, cfg=80)
        pc.w = torch.empty_like(pc.w)
        _lib.check(L.egn_wino4_pack_weight_f32(_lib.ptr(wd), cout, cin, 0, _lib.ptr(pc.w), st), 'wino4 pack')
        y_dev = ops.conv2d_nhwc(x, pc, cin, 1, 1, 0, None, cfg=80)
        torch.cuda.synchronize()
        # (filters that differ in the last bit of a few U values; the output transform amplifies by up to ~100)
        assert (y_dev - y_host).abs().max().item() < 2e-4


@pytest.mark.parametrize('cfg', [82, 83])
@pytest.mark.parametrize('n,cin,cout,res,act', [
    (4, 32, 48, True, 1),       # one region, one stage per half (83) / two stages (82)
    (8, 384, 384, True, 1),     # the 8 x 8 maps of the 384-channel branch (hrnet.py stage 4): 8 co-tiles, 24 stages
    (6, 96, 96, False, 0),      # ragged last region (images 4, 5 + two absent ones), no activation, no residual
    (1, 64, 144, True, 1),      # a single image: three absent images in the region
    (37, 96, 48, True, 0),      # odd batch: 10 regions, the last one with a single image
    (64, 384, 384, True, 1),    # BASELINE configs[1]'s batch: 16 regions x 8 co-tiles (x 2 halves = one item per CU)
])
def test_conv_wino4c_kernel(cfg, n, cin, cout, res, act):
    """Configs 82 / 83, conv_wino4c_kernel<0, 1 | 2>: the F(4x4,3x3) body of csrc/conv_wino4.hip on regions of FOUR
    8 x 8 images (hrnet.py's 384-channel branch), 83 with the input channels of an item split over two blocks that add
    their raw outputs into a zeroed y (conv_wino4_finish_kernel applies scale / shift / residual / ReLU).  Same filter
    pack, oracle and tolerance as config 70; two addends onto zero: 83 is run twice and must be bit-identical."""
    import ctypes as C
    from egonet_amd import _lib, ops
    L = _lib.lib()
    assert L.egn_conv_config_kind(cfg) == 3
    out = (C.c_int * 12)()
    assert L.egn_conv_plan_query(n, 8, 8, cin, cin, cout, cout, 3, 3, 1, 1, 0, cfg, out) == 0
    assert list(out)[5:8] == [8, 8, 4]
    err = _conv_case(n, 8, 8, cin, cout, 3, 1, 1, act=act, use_res=res, cfg=cfg, seed=n + cin + cfg)
    assert err < 5e-4, err
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, 8, 8, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    pc = ops.PackedConv(wt, None, None, kind=3)
    r = torch.randn(n, 8, 8, cout, generator=g).cuda() if res else None
    ya = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=cfg)
    yb = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=cfg)
    yc = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=82)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
    # the split changes the summation order of the two channel halves only (same tolerance as against the oracle)
    assert (ya - yc).abs().max().item() < 5e-4
    for hh, ww, ci in ((16, 16, 48), (8, 16, 48), (4, 4, 48)):
        assert L.egn_conv_plan_query(2, hh, ww, ci, ci, 48, 48, 3, 3, 1, 1, 0, cfg, out) != 0
    if cfg == 83:       # whole 16-channel stages per half
        assert L.egn_conv_plan_query(2, 8, 8, 48, 48, 48, 48, 3, 3, 1, 1, 0, 83, out) != 0


@pytest.mark.parametrize('n,cin,cout,res,act', [(64, 384, 384, True, 1), (10, 96, 96, True, 1), (5, 64, 48, False, 0),
                                                (16, 64, 192, True, 1)])     # (4 co-tiles: the halves of a pair on two XCDs)
def test_conv_wino4c_ticket_path_of_programs(n, cin, cout, res, act):
    """Config 83 inside a program (egn_program_add_conv2d): the op owns one zeroed ticket word per item pair and the layer
    is ONE launch -- the half of a pair that finishes second adds the first one's raw share (read back from y) and applies
    the epilogue -- instead of memset + atomic adds + conv_wino4_finish_kernel (egn_conv2d_f32, no ticket words).  Same
    arithmetic per output: bit-identical to that path; launched many times (every launch must leave the words zero), run
    eagerly and as a captured graph."""
    from egonet_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(n + cin)
    x = torch.randn(n, 8, 8, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    bn = _bn(cout, g)
    pc = ops.PackedConv(wt, None, bn, kind=3)
    r = torch.randn(n, 8, 8, cout, generator=g).cuda() if res else None
    want = ops.conv2d_nhwc(x, pc, cin, 1, 1, act, r, cfg=83)
    y = torch.full((n, 8, 8, cout), float('nan'), device='cuda')
    h = L.egn_program_create(8)
    assert h
    try:
        tensors = [x, pc.w, pc.scale, pc.shift, r, y]
        refs = []
        for slot, t in enumerate(tensors):
            if t is None:
                refs.append(_lib.NULL_REF)
                continue
            assert L.egn_program_bind(h, slot, _lib.ptr(t)) == 0
            refs.append(_lib.Ref(slot, 0))
        assert L.egn_program_add_conv2d(h, *refs, n, 8, 8, cin, cin, cout, cout, 3, 3, 1, 1, act, 0, 83) == 0
        side = torch.cuda.Stream()          # (a capture needs a stream of its own)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            st = _lib.current_stream()
            for _ in range(20):
                y.fill_(float('nan'))
                assert L.egn_program_run(h, st) == 0
                torch.cuda.synchronize()
                assert torch.equal(y, want)
            assert L.egn_program_capture(h, st) == 0
            for _ in range(5):
                y.fill_(float('nan'))
                assert L.egn_program_replay(h, st) == 0
                torch.cuda.synchronize()
                assert torch.equal(y, want)
    finally:
        L.egn_program_destroy(h)


def test_k_split_wait_that_runs_out_fails_the_next_run_of_the_program():
    """[round 6, ADVICE r5] A ticket word that is not zero at launch (a caller that shares words between streams, a
    program run beside itself) makes both halves of an item pair wait for a count that never comes; the wait is bounded
    (2^22 polls), the block raises the owner's error word (word 0) and goes on with invalid output.  Nothing used to read
    that word.  Now: behind every run the program ORs its K-split ops' error words into a pinned mirror, and the NEXT run
    (eager, timed or replayed) fails with EGN_E_STATE after zeroing every word -- the run after that is correct again."""
    from egonet_amd import _lib, ops
    L = _lib.lib()
    n, cin, cout = 4, 64, 48
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 8, 8, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    pc = ops.PackedConv(wt, None, _bn(cout, g), kind=3)
    want = ops.conv2d_nhwc(x, pc, cin, 1, 1, 1, None, cfg=83)
    y = torch.full((n, 8, 8, cout), float('nan'), device='cuda')
    h = L.egn_program_create(8)
    assert h
    try:
        refs = []
        for slot, t in enumerate([x, pc.w, pc.scale, pc.shift, None, y]):
            if t is None:
                refs.append(_lib.NULL_REF)
                continue
            assert L.egn_program_bind(h, slot, _lib.ptr(t)) == 0
            refs.append(_lib.Ref(slot, 0))
        assert L.egn_program_add_conv2d(h, *refs, n, 8, 8, cin, cin, cout, cout, 3, 3, 1, 1, 1, 0, 83) == 0
        assert L.egn_program_ticket_ops(h) == 1
        assert L.egn_program_poke_ticket(h, 1, 0, 1) != 0 and L.egn_program_poke_ticket(h, 0, 1 << 20, 1) != 0
        st = _lib.current_stream()
        for _ in range(3):
            assert L.egn_program_run(h, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(y, want)
        assert L.egn_program_poke_ticket(h, 0, 1, 5) == 0       # the first item pair's word: both halves will wait in vain
        assert L.egn_program_run(h, st) == 0                    # (issued fine; its output is invalid)
        torch.cuda.synchronize()
        assert L.egn_program_run(h, st) == -3                   # EGN_E_STATE: the previous run raised the error word
        torch.cuda.synchronize()
        y.fill_(float('nan'))
        assert L.egn_program_run(h, st) == 0                    # words zeroed by the failing call: healthy again
        torch.cuda.synchronize()
        assert torch.equal(y, want)
        assert L.egn_program_run(h, st) == 0
    finally:
        L.egn_program_destroy(h)


@pytest.mark.parametrize('h,c,cfg', [(64, 48, 51), (32, 96, 51), (16, 192, 51), (8, 384, 56), (16, 192, 57)])
def test_winograd_at_the_bench_batch_size_agrees_with_direct_and_is_linear(h, c, cfg):
    """BASELINE configs[1] size (64 crops): the Winograd kernel of each shape class against the direct
    kernel on the same tensors (two independent HIP implementations of the same layer), and the
    size-independent property of a convolution without activation: conv(x1 + x2) = conv(x1) + conv(x2)."""
    from egonet_amd import _lib, engine
    L = _lib.lib()
    st = _lib.current_stream()
    n = 64
    g = torch.Generator().manual_seed(h + c)
    x1 = torch.randn(n, h, h, c, generator=g).cuda()
    x2 = torch.randn(n, h, h, c, generator=g).cuda()
    wt = torch.randn(c, c, 3, 3, generator=g) / (3 * c ** 0.5)
    wd, wu = engine.pack_conv_weight(wt).cuda(), engine.pack_wino_weight(wt).cuda()
    ones, zeros = torch.ones(c + 16).cuda(), torch.zeros(c + 16).cuda()

    def conv(x, wp, cfg_, act=0):
        y = torch.empty(n, h, h, c, device='cuda')
        _lib.check(L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(ones), _lib.ptr(zeros), None, _lib.ptr(y),
                                    n, h, h, c, c, c, c, 3, 3, 1, 1, act, 0, cfg_, st))
        return y
    yw, yd = conv(x1, wu, cfg), conv(x1, wd, 0)
    scale = float(yd.abs().max())
    assert float((yw - yd).abs().max()) < 3e-5 * max(scale, 1.0)
    lin = conv(x1 + x2, wu, cfg) - (yw + conv(x2, wu, cfg))
    assert float(lin.abs().max()) < 3e-5 * max(scale, 1.0)
    # every output element is written (no stale NaN), ReLU variant clamps exactly at zero
    yr = conv(x1, wu, cfg, act=1)
    assert torch.equal(yr, torch.clamp_min(yw, 0.0))


def test_winograd_filter_pack_device_vs_host():
    """egn_wino_pack_weight_f32 == engine.pack_wino_weight (the float64 host transform the inference
    programs use), forward and data-gradient filters."""
    from egonet_amd import _lib, engine
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    for cout, cin in ((48, 48), (96, 32), (144, 192), (64, 64), (32, 128)):
        wt = torch.randn(cout, cin, 3, 3, generator=g)
        for dgrad in (0, 1):
            if L.egn_wino_weight_floats(cout, cin, dgrad) == 0:
                continue
            dst = torch.empty(L.egn_wino_weight_floats(cout, cin, dgrad), device='cuda')
            _lib.check(L.egn_wino_pack_weight_f32(_lib.ptr(wt.cuda()), cout, cin, dgrad, _lib.ptr(dst),
                                                  _lib.current_stream()))
            src = wt.flip(2, 3).permute(1, 0, 2, 3).contiguous() if dgrad else wt
            want = engine.pack_wino_weight(src)
            assert dst.numel() == want.numel()
            d = (dst.cpu() - want).abs().max().item()
            assert d <= 2e-7 * wt.abs().max().item(), (cout, cin, dgrad, d)


@pytest.mark.parametrize('n,h,w,res,act', [
    (2, 64, 64, True, 1),        # the HRNet shape: 64 tiles
    (1, 8, 16, False, 0),        # exactly one tile, every halo side is padding
    (3, 19, 13, True, 1),        # partial tiles in x and y
    (70, 64, 64, True, 1),       # 2240 tiles > 8 per persistent block: the double-buffered halo pipeline
    (5, 40, 72, False, 1),
])
def test_conv_c48_resident_filter_kernel(n, h, w, res, act):
    """Configs 42 / 44 (csrc/conv_c48.hip): 48 -> 48 3x3, filter resident in LDS, persistent (8 waves x 1 row; 8 x 2 on a
    16-row tile).  41 (4 waves x 2 rows) and 43 (filter in registers) were never selected: probe builds only."""
    import ctypes as C
    from egonet_amd import _lib
    L = _lib.lib()
    assert L.egn_conv_num_configs() >= 44
    probes = bool(L.egn_probe_build())
    out = (C.c_int * 12)()
    for cfg in (41, 42, 43, 44):
        if cfg in (41, 43) and not probes:
            assert L.egn_conv_plan_query(n, h, w, 48, 48, 48, 48, 3, 3, 1, 1, 0, cfg, out) != 0
            continue
        err = _conv_case(n, h, w, 48, 48, 3, 1, 1, act=act, use_res=res, cfg=cfg, seed=n + h)
        assert err < 2e-4, (cfg, err)
    assert L.egn_conv_plan_query(2, 64, 64, 48, 48, 96, 96, 3, 3, 1, 1, 0, 42, out) != 0      # only 48 -> 48
    assert L.egn_conv_plan_query(2, 64, 64, 48, 48, 48, 48, 3, 3, 2, 1, 0, 42, out) != 0      # only stride 1
    buf = C.create_string_buffer(128)
    assert L.egn_conv_config_name(42, buf, 128) == 0 and b'conv_c48_kernel' in buf.value


def test_conv_heads_and_linear():
    assert _conv_case(2, 64, 64, 48, 33, 1, 1, 0, act=0, nchw=True, bias=True, seed=1) < 2e-4
    assert _conv_case(5, 4, 4, 66, 66, 4, 1, 0, act=2, nchw=True, bias=True, seed=2) < 2e-5     # 4x4 valid + sigmoid
    assert _conv_case(64, 1, 1, 66, 1024, 1, 1, 0, act=1, bias=True, seed=4) < 2e-4             # Linear 66->1024
    assert _conv_case(64, 1, 1, 1024, 1024, 1, 1, 0, act=0x11, use_res=True, bias=True, seed=5) < 3e-4
    assert _conv_case(7, 1, 1, 1024, 96, 1, 1, 0, act=0, nchw=True, bias=True, seed=6) < 3e-4
    assert _conv_case(1, 1, 1, 10, 12, 1, 1, 0, act=3, bias=True, seed=7) < 2e-5                # leaky, N=1


def test_fuse_sum_relu_and_layouts():
    from egonet_amd import ops
    g = torch.Generator().manual_seed(3)
    n, c = 2, 48
    t0 = torch.randn(n, c, 32, 32, generator=g)
    t1 = torch.randn(n, c, 16, 16, generator=g)
    t2 = torch.randn(n, c, 8, 8, generator=g)
    t3 = torch.randn(n, c, 4, 4, generator=g)
    ref = F.relu(((t0 + F.interpolate(t1, scale_factor=2, mode='nearest'))
                  + F.interpolate(t2, scale_factor=4, mode='nearest'))
                 + F.interpolate(t3, scale_factor=8, mode='nearest'))
    d = [ops.nchw_to_nhwc(t.cuda()) for t in (t0, t1, t2, t3)]
    y = ops.fuse_sum_relu(d, [0, 1, 2, 3], c)
    assert torch.equal(ops.nhwc_to_nchw(y, c).cpu(), ref)          # same association order -> bit exact
    # term order with the identity in the middle (branch 1 of a 3-branch module)
    ref = F.relu((t1 + t1 * 2) + F.interpolate(t2, scale_factor=2, mode='nearest'))
    y = ops.fuse_sum_relu([d[1], ops.nchw_to_nhwc((t1 * 2).cuda()), d[2]], [0, 0, 1], c)
    assert torch.equal(ops.nhwc_to_nchw(y, c).cpu(), ref)
    # layout round trip with channel padding (35 -> 36) and pad zeroing
    x = torch.randn(3, 35, 5, 7, generator=g)
    xn = ops.nchw_to_nhwc(x.cuda())
    assert xn.shape == (3, 5, 7, 36) and float(xn[..., 35].abs().max()) == 0.0
    assert torch.equal(ops.nhwc_to_nchw(xn, 35).cpu(), x)
    # coordinate ramps == the reference's linspace maps (hrnet.py:461-467)
    from oracle.hrnet_oracle import coordinate_ramps
    buf = torch.zeros(2, 64, 48, 36, device='cuda')
    ops.fill_coord_ramps(buf, 33)
    ramps = coordinate_ramps(48, 64)
    assert torch.equal(buf[..., 33:35].permute(0, 3, 1, 2).cpu(), ramps.expand(2, -1, -1, -1))


def test_decode_kernels_edge_cases_and_golden():
    from egonet_amd.common import img_proc
    g = golden('decode.npz')
    hm = torch.from_numpy(g['hm']).cuda()
    xy, mx, idx = img_proc.hard_arg_max(hm)
    assert np.array_equal(xy.cpu().numpy(), g['hard_preds'])        # masked / tie / all-equal maps
    assert np.array_equal(mx.cpu().numpy(), g['hard_maxvals'])
    want_idx, _ = decode_oracle.argmax_index(g['hm'])
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    sxy, smx = img_proc.soft_arg_max(hm)
    np.testing.assert_allclose(sxy.cpu().numpy(), g['soft_preds'], rtol=0, atol=1e-4)
    assert np.array_equal(smx.cpu().numpy(), g['soft_maxvals'])
    # numpy contract of get_max_preds
    p, m = img_proc.get_max_preds(g['hm'])
    assert isinstance(p, np.ndarray) and np.array_equal(p, g['hard_preds'])
    # large random batch vs the oracle (full-size maps)
    rng = np.random.RandomState(0)
    big = (rng.randn(64, 33, 64, 64) * 4).astype(np.float32)
    t = torch.from_numpy(big).cuda()
    xy, mx, idx = img_proc.hard_arg_max(t)
    oi, om = decode_oracle.argmax_index(big)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(mx.cpu().numpy(), om)
    sxy, _ = img_proc.soft_arg_max(t)
    want, _ = decode_oracle.soft_arg_max(big)
    np.testing.assert_allclose(sxy.cpu().numpy(), want, rtol=0, atol=1e-3)
    # empty batch
    e = torch.zeros(0, 33, 64, 64, device='cuda')
    xy, mx, idx = img_proc.hard_arg_max(e)
    assert xy.shape == (0, 33, 2)


def test_geometry_kernels_vs_fixtures():
    import ctypes as C
    from egonet_amd import _lib
    L = _lib.lib()
    g = golden('pose_solve.npz')
    n = len(g['preds'])
    p = torch.from_numpy(g['preds'].reshape(n, -1)).cuda()
    kx = torch.from_numpy(g['kpts_x']).cuda()
    e = torch.empty(n, 3, dtype=torch.float64, device='cuda')
    a = torch.empty(n, dtype=torch.float64, device='cuda')
    K = g['K']
    st = _lib.current_stream()
    _lib.check(L.egn_pose_solve_f64(_lib.ptr(p), n, _lib.ptr(kx), K[0, 0], K[0, 2], 0, _lib.ptr(e), _lib.ptr(a), st))
    np.testing.assert_allclose(e.cpu().numpy(), g['euler'], atol=1e-9)
    np.testing.assert_allclose(a.cpu().numpy(), g['alpha_proj'], atol=1e-9)
    _lib.check(L.egn_pose_solve_f64(_lib.ptr(p), n, None, 1.0, 0.0, 1, _lib.ptr(e), _lib.ptr(a), st))
    np.testing.assert_allclose(a.cpu().numpy(), g['alpha_trans'], atol=1e-9)
    assert L.egn_pose_solve_f64(_lib.ptr(p), n, None, 1.0, 0.0, 0, _lib.ptr(e), _lib.ptr(a), st) == -1
    # crop -> screen -> normalise
    from egonet_amd import synth
    boxes = synth.synth_boxes(50, seed=4)
    rets = [geometry_oracle.modify_bbox(b, 1.0) for b in boxes]
    c = np.stack([r['c'] for r in rets])
    s = np.stack([r['s'] for r in rets])
    rng = np.random.RandomState(1)
    local = rng.uniform(0, 1, (50, 33, 2)).astype(np.float32)
    stats = synth.synth_lifter_stats()
    scr = torch.empty(50, 66, dtype=torch.float64, device='cuda')
    lin = torch.zeros(50, 68, dtype=torch.float32, device='cuda')
    d = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).cuda()
    ld, cd, sd, mi, si = torch.from_numpy(local).cuda(), d(c), d(s), d(stats['mean_in'].ravel()), d(stats['std_in'].ravel())
    _lib.check(L.egn_keypoints_to_screen_f64(_lib.ptr(ld), 50, 33, 256.0, 256.0, _lib.ptr(cd), _lib.ptr(sd), 256, 256,
                                             _lib.ptr(scr), _lib.ptr(mi), _lib.ptr(si), _lib.ptr(lin), 68, st))
    want = np.stack([geometry_oracle.crop_to_screen((local[i] * 256).astype(np.float32), c[i], s[i], (256, 256)).reshape(-1)
                     for i in range(50)])
    np.testing.assert_allclose(scr.cpu().numpy(), want, rtol=0, atol=1e-9)
    want_in = ((want - stats['mean_in']) / stats['std_in']).astype(np.float32)
    np.testing.assert_allclose(lin.cpu().numpy()[:, :66], want_in, rtol=0, atol=1e-6)
    assert float(lin[:, 66:].abs().max()) == 0.0


@pytest.mark.parametrize('m,fused,use_res,relu1', [
    (32 * 5, True, True, 1),            # fewer tiles than blocks
    (32 * 777, True, True, 1),          # several tiles per block, ragged last round
    (32 * 1201, True, False, 1),        # no residual: the item end only writes
    (32 * 600, False, False, 0),        # the downsample conv: one product, no residual, no ReLU
    (32 * 2048, False, True, 1),        # the last block's conv3
    (64 * 64 * 16, True, True, 1),      # 16 crops of layer1
])
def test_pw_pair_kernel(m, fused, use_res, relu1):
    """[round 5] csrc/conv_pw.hip through egn_pw_pair_f32: layer1's conv3 (+ residual) + ReLU and the next block's conv1 +
    ReLU (libs/model/heatmapModel/hrnet.py:95-133) against torch's fp32 CPU convolutions + eval-mode BatchNorm; the
    BatchNorm scales are folded into the filters as the engine does (engine.fold_pw)."""
    from egonet_amd import _lib, engine
    L = _lib.lib()
    g = torch.Generator().manual_seed(m + 7 * fused + use_res)
    h = torch.randn(m, 64, generator=g)
    res = torch.randn(m, 256, generator=g) if use_res else None
    w3 = (torch.rand(256, 64, 1, 1, generator=g) * 2 - 1) / 8 * 1.7
    w1 = (torch.rand(64, 256, 1, 1, generator=g) * 2 - 1) / 16 * 1.7
    bn3, bn1 = _bn(256, g), _bn(64, g)
    with torch.no_grad():
        want = bn3(F.conv2d(h.t().reshape(1, 64, m, 1), w3)).reshape(256, m).t()
        if res is not None:
            want = want + res
        if relu1:
            want = F.relu(want)
        want_n = F.relu(bn1(F.conv2d(want.t().reshape(1, 256, m, 1), w1))).reshape(64, m).t() if fused else None
    f3, s3 = engine.fold_pw(w3, bn3)
    f1, s1 = engine.fold_pw(w1, bn1)
    d = lambda t: None if t is None else t.contiguous().cuda()             # noqa: E731
    out = torch.full((m, 256), float('nan'), device='cuda')
    hn = torch.full((m, 64), float('nan'), device='cuda') if fused else None
    hd, rd, f3d, s3d, f1d, s1d = d(h), d(res), d(f3), d(s3), d(f1), d(s1)
    for _ in range(2):
        _lib.check(L.egn_pw_pair_f32(_lib.ptr(hd), _lib.ptr(rd), _lib.ptr(f3d), _lib.ptr(s3d),
                                     _lib.ptr(f1d) if fused else None, _lib.ptr(s1d) if fused else None, _lib.ptr(out),
                                     _lib.ptr(hn), m, relu1, _lib.current_stream()), 'pw pair')
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert float((out.cpu() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    if fused:
        assert torch.isfinite(hn).all()
        assert float((hn.cpu() - want_n).abs().max()) < 3e-5 * max(1.0, float(want_n.abs().max()))
    # refused: ragged M, in-place forms, half a second product
    assert L.egn_pw_pair_f32(_lib.ptr(hd), None, _lib.ptr(f3d), _lib.ptr(s3d), None, None, _lib.ptr(out), None, m + 8, 1,
                             _lib.current_stream()) != 0
    assert L.egn_pw_pair_f32(_lib.ptr(hd), _lib.ptr(out), _lib.ptr(f3d), _lib.ptr(s3d), None, None, _lib.ptr(out), None, m, 1,
                             _lib.current_stream()) != 0
    assert L.egn_pw_pair_f32(_lib.ptr(hd), None, _lib.ptr(f3d), _lib.ptr(s3d), _lib.ptr(f1d), _lib.ptr(s1d), _lib.ptr(out),
                             None, m, 1, _lib.current_stream()) != 0


@pytest.mark.parametrize('n,h,w,cout,act', [
    (2, 64, 64, 48, 1),          # 48 -> 48 @ 32 x 32 (fuse down path, first step)
    (2, 64, 64, 96, 0),          # 48 -> 96 @ 32 x 32 (last step of a chain: no ReLU), two co-groups
    (3, 32, 32, 192, 1),         # 48 -> 192 @ 16 x 16
    (5, 16, 16, 384, 0),         # 48 -> 384 @ 8 x 8: more co-groups than a tile row, odd batch
    (1, 4, 16, 48, 1),           # one tile: every tap row / column touches the border
    (70, 16, 32, 96, 1),         # more items than blocks
])
def test_conv_s2r_kernel(n, h, w, cout, act):
    """[round 5] csrc/conv_s2r.hip (cfg 85): 3x3 stride-2 convolution from the 48-channel branch with the filter slice in
    registers and the im2col gather by LDS-DMA, through egn_conv2d_f32 with the direct-packed filter, against torch's fp32
    CPU conv + eval-mode BatchNorm (+ ReLU); refused where it does not apply."""
    import ctypes as C
    from egonet_amd import _lib
    L = _lib.lib()
    out = (C.c_int * 12)()
    assert L.egn_conv_config_kind(85) == 0
    assert L.egn_conv_plan_query(n, h, w, 48, 48, cout, cout, 3, 3, 2, 1, 0, 85, out) == 0
    err = _conv_case(n, h, w, 48, cout, 3, 2, 1, act=act, cfg=85, seed=cout + n)
    assert err < 2e-4, err
    for bad in ((n, h, w, 96, 96, cout, cout, 3, 3, 2, 1), (n, h, w, 48, 48, cout, cout, 3, 3, 1, 1),
                (n, h, w, 48, 48, 64, 64, 3, 3, 2, 1), (n, h, 12, 48, 48, cout, cout, 3, 3, 2, 1)):
        assert L.egn_conv_plan_query(*bad, 0, 85, out) != 0
