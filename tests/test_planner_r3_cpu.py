"""Planner rules of the round-3 kernel families, checked on the host (egn_conv_plan_query needs no GPU): which
shapes the stem kernel (cfg 64), the Winograd F(4x4,3x3) kernel (cfg 70) and the row-GEMM kernel (cfg 79) accept,
which filter packing each expects, and that the shipped table only names configurations that plan for their shape."""
import ctypes as C
import json
import os
import re

from egonet_amd import _lib, tuner

KEY = re.compile(r'n(\d+)_h(\d+)_w(\d+)_ci(\d+)\.(\d+)_co(\d+)\.(\d+)_k(\d+)x(\d+)_s(\d+)_p(\d+)_r(\d+)_o(\d+)$')


def _plans(L, n, h, w, cin, cs_in, cout, cs_out, k, s, p, nchw, cfg):
    out = (C.c_int * 12)()
    return L.egn_conv_plan_query(n, h, w, cin, cs_in, cout, cs_out, k, k, s, p, nchw, cfg, out) == 0


def test_kinds_and_names():
    L = _lib.lib()
    probes = bool(L.egn_probe_build())
    # 65 (first F(4x4,3x3) kernel) and 67 (two 4-wave blocks per CU) were measured and retired: probe builds only
    assert [L.egn_conv_config_kind(c) for c in (59, 64, 65, 67, 70, 79, 80)] == \
        ([1, 0, 2, 1, 3, 0, 3] if probes else [1, 0, -1, -1, 3, 0, 3])
    assert all(L.egn_conv_config_kind(c) == -1 for c in (58, 63, 66, 69, 71, 72, 73, 74, 75, 76, 77, 78, 81))   # stamps / ablations
    if not probes:
        # ... and the product library neither plans nor launches them
        for cfg in (41, 43, 45, 46, 47, 53, 58, 63, 65, 66, 67, 68, 69, 71, 72, 76, 78, 81):
            assert not _plans(L, 64, 64, 64, 48, 48, 48, 48, 3, 1, 1, 0, cfg), cfg
            assert L.egn_conv2d_f32(None, None, None, None, None, None, 64, 64, 64, 48, 48, 48, 48, 3, 3, 1, 1, 1, 0, cfg,
                                    None) != 0, cfg
    buf = C.create_string_buffer(128)
    for cfg, sym in ((64, 'conv_stem_kernel'), (70, 'conv_wino4_kernel<0>'), (80, 'conv_wino4b_kernel<0>'), (79, 'conv_fc_kernel'),
                     (59, 'conv_wino9_kernel<16, 16, 1, 8, 3, 0, 4>'), (67, 'conv_wino9_kernel<8, 16, 1, 4, 3, 0, 2>')):
        assert L.egn_conv_config_name(cfg, buf, 128) == 0 and sym in buf.value.decode(), (cfg, buf.value)


def test_stem_wino4_and_fc_rules():
    L = _lib.lib()
    # stem: 3x3 s2 p1, 3 (4) -> 64 channels, even maps
    assert _plans(L, 2, 256, 256, 3, 4, 64, 64, 3, 2, 1, 0, 64)
    assert not _plans(L, 2, 256, 256, 3, 4, 48, 48, 3, 2, 1, 0, 64)
    assert not _plans(L, 2, 256, 256, 16, 16, 64, 64, 3, 2, 1, 0, 64)
    assert not _plans(L, 2, 256, 256, 3, 4, 64, 64, 3, 1, 1, 0, 64)
    # F(4x4,3x3): 3x3 s1 p1, Cin % 16, Cout % 48, whole 16 x 32 regions
    assert _plans(L, 64, 64, 64, 48, 48, 48, 48, 3, 1, 1, 0, 70)
    assert _plans(L, 5, 32, 32, 96, 96, 144, 144, 3, 1, 1, 0, 70)
    for bad in ((64, 16, 16, 192, 192, 192, 192, 3, 1, 1), (64, 64, 64, 64, 64, 64, 64, 3, 1, 1),
                (64, 64, 64, 24, 24, 48, 48, 3, 1, 1), (64, 64, 64, 48, 48, 48, 48, 3, 2, 1),
                (64, 64, 64, 48, 52, 48, 48, 3, 1, 1), (64, 64, 64, 48, 48, 48, 48, 1, 1, 0)):
        assert not _plans(L, *bad, 0, 70), bad
    assert L.egn_wino4_weight_floats(96, 48) == 2 * 6 * 2 * 12 * 3 * 64 * 4
    # ... conv_wino4b_kernel (cfg 80): whole 16 x 16 regions -- the 16 x 16 maps of the 192-channel branch plan
    assert _plans(L, 64, 16, 16, 192, 192, 192, 192, 3, 1, 1, 0, 80)
    assert _plans(L, 16, 64, 64, 48, 48, 48, 48, 3, 1, 1, 0, 80) and _plans(L, 3, 32, 48, 16, 16, 96, 96, 3, 1, 1, 0, 80)
    for bad in ((64, 8, 8, 384, 384, 384, 384, 3, 1, 1), (64, 24, 16, 48, 48, 48, 48, 3, 1, 1),
                (64, 16, 16, 24, 24, 48, 48, 3, 1, 1), (64, 16, 16, 192, 192, 64, 64, 3, 1, 1),
                (64, 16, 16, 192, 192, 192, 192, 3, 2, 1)):
        assert not _plans(L, *bad, 0, 80), bad
    out = (C.c_int * 12)()
    assert L.egn_conv_plan_query(64, 16, 16, 192, 192, 192, 192, 3, 3, 1, 1, 0, 80, out) == 0
    assert list(out)[5:8] == [16, 16, 1] and out[10] * out[11] == 64 * 4      # one region per image x 4 co-tiles = 256 items
    # row GEMM: 1x1 s1 p0, Cin % 16, Cout % 16; NCHW output only for 1 x 1 maps
    assert _plans(L, 64, 1, 1, 1024, 1024, 1024, 1024, 1, 1, 0, 0, 79)
    assert _plans(L, 64, 1, 1, 1024, 1024, 96, 96, 1, 1, 0, 1, 79)
    assert _plans(L, 64, 8, 8, 384, 384, 48, 48, 1, 1, 0, 0, 79)
    assert not _plans(L, 64, 8, 8, 384, 384, 48, 48, 1, 1, 0, 1, 79)
    assert not _plans(L, 64, 1, 1, 66, 68, 1024, 1024, 1, 1, 0, 0, 79)
    assert not _plans(L, 64, 64, 64, 48, 48, 33, 33, 1, 1, 0, 0, 79)


def test_every_table_entry_plans_for_its_shape():
    L = _lib.lib()
    with open(tuner.TABLE_PATH) as f:
        table = json.load(f)
    n70 = n79 = 0
    for key, ent in table.items():
        m = KEY.match(key)
        assert m, key
        n, h, w, cin, cs_in, cout, cs_out, kh, kw, s, p, r, o = (int(v) for v in m.groups())
        cfg = int(ent['cfg'])
        if cfg <= 0:
            continue
        assert L.egn_conv_config_kind(cfg) >= 0, (key, cfg)
        out = (C.c_int * 12)()
        assert L.egn_conv_plan_query(n, h, w, cin, cs_in, cout, cs_out, kh, kw, s, p, o, cfg, out) == 0, (key, cfg)
        n70 += cfg == 70
        n79 += cfg == 79
        # a caller that cannot feed Winograd filters still gets a configuration it can run
        alt = tuner._pick(ent, allow_wino=False, allow_f43=False)
        assert alt == 0 or L.egn_conv_config_kind(alt) == 0, (key, alt)
    assert n70 >= 8 and n79 >= 15, (n70, n79)


def test_row_gemm_kernel_reads_the_standard_filter_pack():
    """conv_fc_kernel's operand addressing restated in numpy: lane (channel n0 + li, k lanes 4 kq .. 4 kq + 3 of chunk c)
    reads the float4 at ((c * 4 + kq) * CoutP + n0 + li) of engine.pack_conv_weight's layout; the four waves take
    chunks c = wave, wave + 4, ... and their partial tiles are summed w0 + w1 + w2 + w3."""
    import numpy as np
    import torch
    from egonet_amd import engine
    rng = np.random.default_rng(3)
    rows, cin, cout = 20, 80, 48
    w = rng.standard_normal((cout, cin, 1, 1)).astype(np.float32)
    x = rng.standard_normal((rows, cin)).astype(np.float32)
    wp = engine.pack_conv_weight(torch.from_numpy(w)).numpy().reshape(cin // 16, 1, 4, cout, 4)   # chunk, tap, quad, co, e
    y = np.zeros((rows, cout), np.float64)
    for wave in range(4):
        part = np.zeros((rows, cout), np.float64)
        for c in range(wave, cin // 16, 4):
            for kq in range(4):
                for e in range(4):
                    k = 16 * c + 4 * kq + e
                    part += np.outer(x[:, k], wp[c, 0, kq, :, e])
        y += part
    np.testing.assert_allclose(y, x.astype(np.float64) @ w[:, :, 0, 0].astype(np.float64).T, rtol=0, atol=1e-5)
