"""CPU checks of the conv kernel DESIGN: the product's weight packing
(egonet_amd.engine.pack_conv_weight / fold_scale_shift), the library's host
tile planner (egn_conv_plan_query) and the lane-level dataflow restated in
tests/conv_emulator.py, against torch.nn.functional.conv2d."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import conv_emulator
from egonet_amd import engine, _lib


def _plan(args, cfg=0):
    out = (C.c_int * 12)()
    rc = _lib.lib().egn_conv_plan_query(*args, cfg, out)
    assert rc == 0, rc
    return list(out)


CASES = [
    # N  H   W  Cin Cout k  s  p  act res  nchw  cfg
    (2, 8, 8, 20, 24, 3, 1, 1, 1, True, False, 0),     # ragged channels, residual, relu
    (1, 9, 7, 16, 16, 3, 2, 1, 0, False, False, 0),    # odd map, stride 2
    (3, 4, 4, 6, 10, 4, 1, 0, 2, False, True, 0),      # 4x4 valid conv + sigmoid, NCHW out
    (2, 6, 10, 35, 7, 1, 1, 0, 0, False, True, 0),     # 1x1 head with bias, NCHW out
    (5, 1, 1, 10, 40, 1, 1, 0, 0x11, True, False, 0),  # Linear with residual-after-act
    (1, 12, 12, 8, 48, 3, 1, 1, 1, False, False, 6),   # forced config (128x48)
    (1, 8, 8, 4, 64, 3, 2, 1, 1, False, False, 8),     # stem-like, forced 64x64
    (2, 8, 8, 20, 24, 3, 1, 1, 1, True, False, 16),    # LDS-DMA family (pixel-major halo tile)
    (1, 9, 7, 16, 16, 3, 2, 1, 0, False, False, 18),   # LDS-DMA family, stride 2
]


@pytest.mark.parametrize('case', CASES)
def test_emulated_kernel_matches_conv2d(case):
    n, h, w, cin, cout, k, s, p, act, use_res, nchw, cfg = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    cs_in = (cin + 3) // 4 * 4
    cs_out = cout if nchw else (cout + 3) // 4 * 4
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    bias = torch.randn(cout, generator=g)
    bn = torch.nn.BatchNorm2d(cout)
    bn.weight.data = 0.5 + torch.rand(cout, generator=g)
    bn.bias.data = torch.randn(cout, generator=g) * 0.2
    bn.running_mean = torch.randn(cout, generator=g) * 0.2
    bn.running_var = 0.5 + torch.rand(cout, generator=g)
    bn.eval()
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    res = torch.randn(n, cout, ho, wo, generator=g) if use_res else None

    with torch.no_grad():
        ref = bn(F.conv2d(x, wt, bias, s, p))
        a = act & 0xf
        f = {0: lambda t: t, 1: F.relu, 2: torch.sigmoid, 3: lambda t: F.leaky_relu(t, 0.01)}[a]
        if res is not None and not (act & 0x10):
            ref = ref + res
        ref = f(ref)
        if res is not None and (act & 0x10):
            ref = res + ref

    x_nhwc = np.zeros((n, h, w, cs_in), np.float32)
    x_nhwc[..., :cin] = x.permute(0, 2, 3, 1).numpy()
    res_nhwc = None
    if res is not None:
        res_nhwc = np.zeros((n, ho, wo, cs_out), np.float32)
        res_nhwc[..., :cout] = res.permute(0, 2, 3, 1).numpy()
    wpack = engine.pack_conv_weight(wt).numpy()
    scale, shift = engine.fold_scale_shift(cout, bias, bn)
    plan = _plan((n, h, w, cin, cs_in, cout, cs_out, k, k, s, p, int(nchw)), cfg)
    y = conv_emulator.emulate(x_nhwc, wpack, scale.numpy(), shift.numpy(), res_nhwc, plan, n, h, w, cin,
                              cs_in, cout, cs_out, k, k, s, p, act, nchw)
    if nchw:
        got = y
    else:
        assert np.all(y[..., cout:] == 0.0), 'pad channels must be written as zeros'
        got = y[..., :cout].transpose(0, 3, 1, 2)
    assert not np.isnan(got).any(), 'every output element must be written exactly by some lane'
    np.testing.assert_allclose(got, ref.numpy(), rtol=0, atol=2e-5)


def test_pack_layout_is_chunk_tap_quad_cout_4():
    wt = torch.arange(2 * 5 * 3 * 3, dtype=torch.float32).reshape(2, 5, 3, 3)
    p = engine.pack_conv_weight(wt).reshape(1, 9, 4, 16, 4)      # [chunk][tap][quad][CoutP][r]
    for co in range(2):
        for ci in range(5):
            for t in range(9):
                assert p[0, t, ci // 4, co, ci % 4] == wt[co, ci, t // 3, t % 3]
    assert p[0, :, :, 2:, :].abs().sum() == 0 and p[0, :, 1, :, 1:].abs().sum() == 0


def test_planner_limits():
    # every HRNet-W48 layer class fits the default 64 KB LDS budget and keeps TW sane
    for args in [(64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 0), (64, 8, 8, 384, 384, 384, 384, 3, 3, 1, 1, 0),
                 (64, 256, 256, 3, 4, 64, 64, 3, 3, 2, 1, 0), (64, 1, 1, 66, 68, 1024, 1024, 1, 1, 1, 0, 0)]:
        plan = _plan(args)
        assert plan[9] <= 64 * 1024
        wm, wn, mt, nt, th, tw, tnb = plan[1:8]
        assert th * tw * tnb == wm * mt * 16
    # bad arguments are reported, not crashed on
    out = (C.c_int * 12)()
    assert _lib.lib().egn_conv_plan_query(1, 8, 8, 6, 6, 8, 8, 3, 3, 1, 1, 0, 0, out) == -1   # cs_in % 4


def test_filter_resident_configs_plan_only_the_48_channel_3x3_layers():
    """Config 42 (csrc/conv_c48.hip): fixed 8x16 tile, filter (82 944 B) + two 48-channel halo buffers
    (2 x 36 864 B) in LDS, refused for every other layer.  41 (four waves) and 43 (four waves that copy the whole
    filter into registers) were measured and never selected: since round 4 they exist in probe builds only
    (-DEGN_PROBES) -- the product library refuses to plan them."""
    L = _lib.lib()
    assert L.egn_conv_num_configs() >= 44
    probes = bool(L.egn_probe_build())
    out = (C.c_int * 12)()
    if not probes:
        for cfg in (41, 43):
            assert L.egn_conv_config_kind(cfg) == -1
            assert L.egn_conv_plan_query(64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 0, cfg, out) != 0
    for cfg, waves in (((41, 4), (42, 8), (43, 4)) if probes else ((42, 8),)):
        plan = _plan((64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 0), cfg=cfg)
        cfg_id, wm, wn, mt, nt, th, tw, tnb, tps, lds = plan[:10]
        assert (cfg_id, wm, wn, mt, nt) == (cfg, waves, 1, 8 // waves, 3)
        assert (th, tw, tnb, tps) == (8, 16, 1, 9) and lds == 82944 + 2 * 36864
        name = C.create_string_buffer(96)
        assert L.egn_conv_config_name(cfg, name, 96) == 0
        assert name.value == (b'conv_c48r_kernel(ConvArgs)' if cfg == 43 else
                              b'void conv_c48_kernel<%d>(ConvArgs)' % waves)
        out = (C.c_int * 12)()
        for bad in ((64, 64, 64, 48, 48, 96, 96, 3, 3, 1, 1, 0),      # 48 -> 96
                    (64, 64, 64, 96, 96, 48, 48, 3, 3, 1, 1, 0),      # 96 -> 48
                    (64, 64, 64, 48, 48, 48, 48, 3, 3, 2, 1, 0),      # stride 2
                    (64, 64, 64, 48, 48, 48, 48, 1, 1, 1, 0, 0),      # 1x1
                    (64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 1)):     # NCHW output
            assert L.egn_conv_plan_query(*bad, cfg, out) != 0
        # ragged maps still plan (partial tiles are masked in the kernel)
        assert _plan((3, 19, 13, 48, 48, 48, 48, 3, 3, 1, 1, 0), cfg=cfg)[5:8] == [8, 16, 1]


def test_tall_tile_filter_resident_config():
    """Config 44: 16 x 16 tile, 8 waves x 2 rows, the halo as a ring of three 16-channel chunks
    (3 x 21 504 B) next to the filter."""
    L = _lib.lib()
    plan = _plan((64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 0), cfg=44)
    cfg_id, wm, wn, mt, nt, th, tw, tnb, tps, lds = plan[:10]
    assert (cfg_id, wm, mt, nt, th, tw, tnb, tps) == (44, 8, 2, 3, 16, 16, 1, 9)
    assert lds == 82944 + 3 * 21504 and lds <= 160 * 1024 - 512
    name = C.create_string_buffer(96)
    assert L.egn_conv_config_name(44, name, 96) == 0 and name.value == b'conv_c48t_kernel(ConvArgs)'
    out = (C.c_int * 12)()
    assert L.egn_conv_plan_query(64, 64, 64, 48, 48, 96, 96, 3, 3, 1, 1, 0, 44, out) != 0
    assert _plan((3, 19, 13, 48, 48, 48, 48, 3, 3, 1, 1, 0), cfg=44)[5:8] == [16, 16, 1]


def test_winograd_configs_plan_and_kinds():
    """Configs 45 / 46 (csrc/conv_wino.hip, the 4-wave kernel: probe builds only since round 4) and 51.. (the
    8-wave kernels): fused Winograd F(2x2,3x3).  Fixed tiles (16 x 16 of one image / four 8 x 8 images), two U slabs
    (2 x 49 152 B) + two quad-plane halo buffers in LDS."""
    L = _lib.lib()
    probes = bool(L.egn_probe_build())
    k4 = 1 if probes else -1
    assert [L.egn_conv_config_kind(c) for c in (0, 1, 44, 45, 46, 47, 51, 52, 53, 999)] == \
        [-1, 0, 0, k4, k4, -1, 1, 1, -1, -1]
    # the 8-wave variants: exact-size halo planes (no padding to whole 512-thread pieces) + the
    # [waves][2][48] double table of the BatchNorm partial sums (training)
    p8 = _plan((64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 0), cfg=51)
    assert p8[0] == 51 and p8[1] == 8 and p8[5:8] == [16, 16, 1] and p8[9] == 2 * 49152 + 2 * 4 * 432 * 16 + 8 * 2 * 48 * 8
    p8 = _plan((64, 8, 8, 384, 384, 384, 384, 3, 3, 1, 1, 0), cfg=52)
    assert p8[5:8] == [8, 8, 4] and p8[9] == 2 * 49152 + 2 * 4 * 448 * 16 + 8 * 2 * 48 * 8 and p8[11] == 8
    # 4 waves on 32 tiles: twice the work items for the 8 x 8 maps (128 -> 256 at batch 64) / small batches
    p4 = _plan((64, 8, 8, 384, 384, 384, 384, 3, 3, 1, 1, 0), cfg=56)
    assert p4[1] == 4 and p4[5:8] == [8, 8, 2] and p4[10] * p4[11] == 256 and p4[9] == 2 * 49152 + 2 * 4 * 256 * 16 + 4 * 2 * 48 * 8
    p4 = _plan((32, 16, 16, 192, 192, 192, 192, 3, 3, 1, 1, 0), cfg=57)
    assert p4[5:8] == [8, 16, 1] and p4[10] * p4[11] == 256
    assert L.egn_conv_plan_query(64, 16, 16, 192, 192, 192, 192, 3, 3, 1, 1, 0, 56, (C.c_int * 12)()) != 0
    if not probes:       # the product library refuses the retired 4-wave family; the rest of this test is about it
        assert L.egn_conv_plan_query(64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 0, 45, (C.c_int * 12)()) != 0
        assert L.egn_conv_plan_query(64, 8, 8, 384, 384, 384, 384, 3, 3, 1, 1, 0, 46, (C.c_int * 12)()) != 0
    for cfg, shape, tile in ((45, (64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 0), [16, 16, 1]),
                             (45, (64, 16, 16, 192, 192, 192, 192, 3, 3, 1, 1, 0), [16, 16, 1]),
                             (46, (64, 8, 8, 384, 384, 384, 384, 3, 3, 1, 1, 0), [8, 8, 4])) if probes else ():
        plan = _plan(shape, cfg=cfg)
        assert plan[0] == cfg and plan[5:8] == tile
        assert plan[9] == 2 * 49152 + 2 * 1792 * 16 and plan[9] <= 160 * 1024 - 512
        assert plan[11] == shape[5] // 48                       # co-tiles
    out = (C.c_int * 12)()
    for bad in ((64, 64, 64, 48, 48, 48, 48, 3, 3, 2, 1, 0),          # stride 2
                (64, 64, 64, 48, 48, 48, 48, 1, 1, 1, 0, 0),          # 1x1
                (64, 64, 64, 35, 36, 48, 48, 3, 3, 1, 1, 0),          # padded / odd input channels
                (64, 64, 64, 48, 48, 64, 64, 3, 3, 1, 1, 0),          # Cout % 48 (the 4-wave kernel has no 32-wide co-tile)
                (64, 63, 64, 48, 48, 48, 48, 3, 3, 1, 1, 0),          # odd map
                (64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 1)):         # NCHW output
        assert L.egn_conv_plan_query(*bad, 45, out) != 0
        if bad[5] != 64:
            assert L.egn_conv_plan_query(*bad, 59, out) != 0      # conv_wino9_kernel: the same rules (64 = 2 x 32 plans)
    assert L.egn_conv_plan_query(64, 16, 16, 192, 192, 192, 192, 3, 3, 1, 1, 0, 46, out) != 0   # 46: 8x8 maps only
    assert L.egn_conv_plan_query(64, 16, 16, 192, 192, 192, 192, 3, 3, 1, 1, 0, 60, out) != 0   # 60 likewise
    name = C.create_string_buffer(96)
    assert L.egn_conv_config_name(45, name, 96) == 0 and name.value == b'void conv_wino_kernel<16, 16, 1, 0>(ConvArgs)'
    assert L.egn_wino_weight_floats(96, 48, 0) == 96 * 48 * 16 and L.egn_wino_weight_floats(40, 48, 0) == 0
    # the 8-wave kernels also take 32-channel co-tiles: W32 widths plan, LDS = 2 x 32 KB U slabs + halo + stats
    p32 = _plan((32, 64, 48, 32, 32, 32, 32, 3, 3, 1, 1, 0), cfg=51)
    assert p32[5:8] == [16, 16, 1] and p32[11] == 1 and p32[9] == 2 * 32768 + 2 * 4 * 432 * 16 + 8 * 2 * 32 * 8
    assert _plan((8, 8, 8, 256, 256, 256, 256, 3, 3, 1, 1, 0), cfg=56)[11] == 8
    assert L.egn_conv_plan_query(8, 16, 16, 80, 80, 80, 80, 3, 3, 1, 1, 0, 51, out) != 0       # 80: neither 48 | nor 32 |
    assert L.egn_wino_weight_floats(64, 64, 0) == 64 * 64 * 16 and L.egn_wino_weight_floats(256, 128, 1) == 256 * 128 * 16


def test_winograd_halo_layout_is_bank_conflict_free():
    """The quad-plane LDS image with skewed rows / images: all 16 lanes of every ds_read_b128 lane
    group of a patch read hit distinct 16-byte columns (the pixel-major image of the direct kernels
    would put 4 of them on one)."""
    import wino_emulator
    assert wino_emulator.worst_bank_conflict(16, 16, 1) == 1
    assert wino_emulator.worst_bank_conflict(8, 8, 4) == 1
    assert wino_emulator.worst_bank_conflict(8, 8, 2) == 1          # the 32-tile forms of the 8-wave kernel
    assert wino_emulator.worst_bank_conflict(8, 16, 1) == 1


def test_winograd_kernel_design_vs_torch():
    """Lane-level emulation of conv_wino.hip (tests/wino_emulator.py) on the product code's filter
    packing: work-item map, halo slots incl. zero padding, fragment mapping, both transforms and the
    epilogue addressing reproduce torch's conv2d (+scale/shift, residual, ReLU); every output is
    written exactly once."""
    import wino_emulator
    assert wino_emulator.conv_case(1, 16, 16, 16, 48, 16, 16, 1) < 2e-5
    assert wino_emulator.conv_case(1, 24, 8, 32, 96, 16, 16, 1, seed=1) < 2e-5     # partial tiles, 2 co-tiles
    assert wino_emulator.conv_case(5, 8, 8, 16, 48, 8, 8, 4, seed=2) < 2e-5        # partial image batch


def test_winograd_frequency_halves_kernel_design_vs_torch():
    """conv_wino8_kernel (what runs): waves = (m-tile, frequency half), the three patch rows a half
    reads, its half of the U slab, the t1 / t2 exchange between the two waves of an m-tile and the
    output rows a = fh each wave stores -- for all four geometries (8 waves on 64 tiles, 4 waves on
    32 tiles), incl. partial tiles / partial image batches."""
    import wino_emulator
    assert wino_emulator.conv_case8(1, 16, 16, 16, 48, 16, 16, 1) < 2e-5
    assert wino_emulator.conv_case8(1, 24, 8, 32, 96, 16, 16, 1, seed=1) < 2e-5
    assert wino_emulator.conv_case8(5, 8, 8, 16, 48, 8, 8, 4, seed=2) < 2e-5
    assert wino_emulator.conv_case8(3, 8, 8, 32, 48, 8, 8, 2, seed=3) < 2e-5       # two images per block, odd batch
    assert wino_emulator.conv_case8(1, 12, 16, 16, 96, 8, 16, 1, seed=4) < 2e-5    # 8 x 16 tile, partial rows
    # 32-channel co-tiles (NT = 2): the W32 / Pedestrian widths and the 64-channel layers
    assert wino_emulator.conv_case8(1, 16, 16, 32, 64, 16, 16, 1, seed=5) < 2e-5
    assert wino_emulator.conv_case8(3, 8, 8, 16, 32, 8, 8, 2, seed=6) < 2e-5
