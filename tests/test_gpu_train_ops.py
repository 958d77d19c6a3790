"""Building blocks of the native training step (csrc/train_ops.hip) through the
C ABI, each against plain torch on the same device tensors copied to the host."""
import numpy as np
import pytest
import torch

from egonet_amd import _lib

pytestmark = pytest.mark.gpu


def _st():
    return _lib.current_stream()


def _ws(cols):
    L = _lib.lib()
    return torch.zeros(L.egn_colreduce_ws_bytes(cols) // 4, device='cuda')


@pytest.mark.parametrize('rows,cols,ld', [(7, 5, 8), (64, 96, 96), (100, 33, 36), (4096, 66, 68)])
def test_transpose(rows, cols, ld):
    L = _lib.lib()
    g = torch.Generator().manual_seed(rows)
    src = torch.randn(rows, ld, generator=g).cuda()
    ldd = (rows + 3) // 4 * 4
    dst = torch.full((cols, ldd), 7.0, device='cuda')
    _lib.check(L.egn_transpose_f32(_lib.ptr(src), rows, cols, ld, _lib.ptr(dst), ldd, _st()))
    want = torch.zeros(cols, ldd)
    want[:, :rows] = src.cpu()[:, :cols].t()
    assert torch.equal(dst.cpu(), want)        # pad columns are zeroed


@pytest.mark.parametrize('cout,cin,transpose', [(12, 10, 0), (12, 10, 1), (1024, 66, 0), (96, 1024, 1), (33, 40, 1)])
def test_pack_matrix_feeds_the_conv_kernel(cout, cin, transpose):
    """pack on the device, then run the 1x1 conv: out = a @ W^T exactly as F.linear."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(cout * 7 + cin)
    w = torch.randn(cout, cin, generator=g) * 0.1
    src = (w.t().contiguous() if transpose else w).cuda()
    ld = src.shape[1]
    coutp, nchunk = (cout + 15) // 16 * 16, (cin + 15) // 16
    wp = torch.zeros(nchunk * 4 * coutp * 4, device='cuda')
    _lib.check(L.egn_pack_matrix_f32(_lib.ptr(src), ld, cout, cin, transpose, _lib.ptr(wp), _st()))
    rows, ld_a = 40, (cin + 3) // 4 * 4
    a = torch.zeros(rows, ld_a)
    a[:, :cin] = torch.randn(rows, cin, generator=g)
    ad = a.cuda()
    out = torch.zeros(rows, cout, device='cuda')
    one, zero = torch.ones(coutp, device='cuda'), torch.zeros(coutp, device='cuda')
    _lib.check(L.egn_conv2d_f32(_lib.ptr(ad), _lib.ptr(wp), _lib.ptr(one), _lib.ptr(zero), None, _lib.ptr(out),
                                rows, 1, 1, cin, ld_a, cout, cout, 1, 1, 1, 0, 0, 1, 0, _st()))
    want = a[:, :cin].double() @ w.double().t()
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=0, atol=1e-5 * max(1.0, float(want.abs().max())))


@pytest.mark.parametrize('cout,cin', [(1024, 66), (96, 1024), (12, 10)])
def test_linear_weight_as_1x1_filter_packs_like_pack_matrix(cout, cin):
    """The lifter step packs its Linear weights with the conv filter packer (one batched launch per step):
    same bits as egn_pack_matrix_f32 for the forward (dgrad 0) and the transposed (dgrad 1) use."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(cout + cin)
    w = torch.randn(cout, cin, generator=g).cuda()
    for dgrad in (0, 1):
        n_out, n_in = (cin, cout) if dgrad else (cout, cin)
        nfl = L.egn_packed_weight_floats(cout, cin, 1, 1, dgrad)
        assert nfl == ((n_in + 15) // 16) * 4 * ((n_out + 15) // 16 * 16) * 4
        a = torch.full((nfl,), 7.0, device='cuda')
        b = torch.full((nfl,), 9.0, device='cuda')
        _lib.check(L.egn_pack_conv_weight_f32(_lib.ptr(w), cout, cin, 1, 1, dgrad, _lib.ptr(a), _st()))
        _lib.check(L.egn_pack_matrix_f32(_lib.ptr(w), cin, n_out, n_in, dgrad, _lib.ptr(b), _st()))
        assert torch.equal(a, b)


@pytest.mark.parametrize('rows,cols', [(4, 12), (16, 128), (250, 96), (4096, 1024)])
def test_column_reductions(rows, cols):
    L = _lib.lib()
    g = torch.Generator().manual_seed(rows + cols)
    z = (torch.randn(rows, cols, generator=g) * 2 + 0.5).cuda()
    ws = _ws(cols)
    s = torch.zeros(cols, device='cuda')
    _lib.check(L.egn_colsum_f32(_lib.ptr(z), rows, cols, cols, _lib.ptr(s), _lib.ptr(ws), _st()))
    zd = z.cpu().double()
    np.testing.assert_allclose(s.cpu().numpy(), zd.sum(0).numpy(), rtol=1e-6, atol=1e-5)
    mean, istd, varu = (torch.zeros(cols, device='cuda') for _ in range(3))
    rm, rv = torch.full((cols,), 0.3, device='cuda'), torch.full((cols,), 2.0, device='cuda')
    _lib.check(L.egn_bn_stats_f32(_lib.ptr(z), rows, cols, cols, 1e-5, _lib.ptr(mean), _lib.ptr(istd),
                                  _lib.ptr(varu), _lib.ptr(rm), _lib.ptr(rv), 0.1, _lib.ptr(ws), _st()))
    np.testing.assert_allclose(rm.cpu().numpy(), 0.9 * 0.3 + 0.1 * zd.mean(0).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), 0.9 * 2.0 + 0.1 * zd.var(0, unbiased=True).numpy(), rtol=2e-6)
    np.testing.assert_allclose(mean.cpu().numpy(), zd.mean(0).numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(istd.cpu().numpy(), (zd.var(0, unbiased=False) + 1e-5).rsqrt().numpy(), rtol=2e-6)
    np.testing.assert_allclose(varu.cpu().numpy(), zd.var(0, unbiased=True).numpy(), rtol=2e-6)


@pytest.mark.parametrize('rows,cols,p,with_res,relu', [
    (16, 128, 0.0, False, 1), (64, 100, 0.5, False, 1), (512, 1024, 0.5, False, 1),
    (2 * 16 * 16, 48, 0.0, True, 1),      # BasicBlock tail: relu(bn2(conv2) + residual)
    (2 * 8 * 8, 96, 0.0, False, 0),       # fuse-layer BatchNorm without ReLU
    (9000, 36, 0.0, True, 1),             # many rows, few columns (NHWC maps)
])
def test_bn_relu_dropout_forward_backward(rows, cols, p, with_res, relu):
    """bn_act_fwd / bn_bwd_sums / bn_bwd_dz against autograd through
    batch_norm(training) (+ residual) -> relu -> mask."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(cols)
    z = torch.randn(rows, cols, generator=g) * 1.5 + 0.2
    gamma = torch.rand(cols, generator=g) + 0.5
    beta = torch.randn(cols, generator=g) * 0.3
    dy = torch.randn(rows, cols, generator=g)
    res = torch.randn(rows, cols, generator=g) if with_res else None
    mask = (torch.rand(rows, cols, generator=g) >= p).float() if p > 0 else None
    keep = 1.0 / (1.0 - p) if p > 0 else 1.0
    # torch (float64 autograd)
    zz, gg, bb = (t.double().requires_grad_(True) for t in (z, gamma, beta))
    rr = res.double().requires_grad_(True) if with_res else None
    pre = torch.nn.functional.batch_norm(zz, None, None, gg, bb, True, 0.1, 1e-5)
    if with_res:
        pre = pre + rr
    y = torch.relu(pre) if relu else pre
    if mask is not None:
        y = y * mask.double() * keep
    y.backward(dy.double())
    # native
    zd, gd, bd, dyd = z.cuda(), gamma.cuda(), beta.cuda(), dy.cuda()
    md = mask.cuda() if mask is not None else None
    rd = res.cuda() if with_res else None
    ws = _ws(cols)
    mean, istd, varu, dbeta, dgamma = (torch.zeros(cols, device='cuda') for _ in range(5))
    _lib.check(L.egn_bn_stats_f32(_lib.ptr(zd), rows, cols, cols, 1e-5, _lib.ptr(mean), _lib.ptr(istd),
                                  _lib.ptr(varu), None, None, 0.0, _lib.ptr(ws), _st()))
    yd = torch.zeros(rows, cols, device='cuda')
    _lib.check(L.egn_bn_act_fwd_f32(_lib.ptr(zd), _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(gd), _lib.ptr(bd),
                                    _lib.ptr(md), keep, relu, _lib.ptr(rd), _lib.ptr(yd), rows, cols, cols, _st()))
    np.testing.assert_allclose(yd.cpu().numpy(), y.detach().numpy(), rtol=0, atol=2e-5)
    _lib.check(L.egn_bn_bwd_sums_f32(_lib.ptr(dyd), _lib.ptr(zd), _lib.ptr(md), keep, _lib.ptr(mean), _lib.ptr(istd),
                                     _lib.ptr(gd), _lib.ptr(bd), relu, _lib.ptr(rd), rows, cols, cols,
                                     _lib.ptr(dbeta), _lib.ptr(dgamma), _lib.ptr(ws), _st()))
    dz = torch.zeros(rows, cols, device='cuda')
    dres = torch.zeros(rows, cols, device='cuda') if with_res else None
    _lib.check(L.egn_bn_bwd_dz_f32(_lib.ptr(dyd), _lib.ptr(zd), _lib.ptr(md), keep, _lib.ptr(mean), _lib.ptr(istd),
                                   _lib.ptr(gd), _lib.ptr(bd), relu, _lib.ptr(rd), _lib.ptr(dbeta), _lib.ptr(dgamma),
                                   _lib.ptr(dz), _lib.ptr(dres), rows, cols, cols, _st()))
    # an element whose pre-activation is within rounding of 0 may flip its ReLU
    # gate between fp32 and fp64; exclude columns that hold one
    safe = (pre.detach().abs() > 1e-5).all(0).numpy()
    assert safe.mean() > 0.8
    tol = 1e-4 * max(1.0, (rows / 512.0) ** 0.5)
    np.testing.assert_allclose(dbeta.cpu().numpy()[safe], bb.grad.numpy()[safe], rtol=1e-5, atol=tol)
    np.testing.assert_allclose(dgamma.cpu().numpy()[safe], gg.grad.numpy()[safe], rtol=1e-5, atol=tol)
    np.testing.assert_allclose(dz.cpu().numpy()[:, safe], zz.grad.numpy()[:, safe], rtol=0, atol=1e-4)
    if with_res:
        np.testing.assert_allclose(dres.cpu().numpy()[:, safe], rr.grad.numpy()[:, safe], rtol=0, atol=1e-6)


def test_mse_loss_and_gradient():
    L = _lib.lib()
    g = torch.Generator().manual_seed(4)
    pred, tgt = torch.randn(300, 96, generator=g), torch.randn(300, 96, generator=g)
    pd, td = pred.cuda(), tgt.cuda()
    dp = torch.zeros_like(pd)
    loss = torch.zeros(1, dtype=torch.float64, device='cuda')
    _lib.check(L.egn_mse_f32(_lib.ptr(pd), _lib.ptr(td), 300, 96, 96, 96, 1.0, 0, _lib.ptr(dp), _lib.ptr(loss), _st()))
    pp = pred.double().requires_grad_(True)
    want = torch.nn.functional.mse_loss(pp, tgt.double())
    want.backward()
    # weight 0.5 + accumulate on top of an existing gradient (heat-map term)
    acc = torch.ones_like(pd)
    loss2 = torch.zeros(1, dtype=torch.float64, device='cuda')
    _lib.check(L.egn_mse_f32(_lib.ptr(pd), _lib.ptr(td), 300, 96, 96, 96, 0.5, 1, _lib.ptr(acc), _lib.ptr(loss2), _st()))
    np.testing.assert_allclose(acc.cpu().numpy(), 1.0 + 0.5 * pp.grad.numpy(), rtol=1e-6, atol=1e-7)
    assert abs(float(loss2.item()) - 0.5 * float(want.detach())) < 1e-9
    assert abs(float(loss.item()) - float(want)) < 1e-9
    np.testing.assert_allclose(dp.cpu().numpy(), pp.grad.numpy(), rtol=1e-6, atol=1e-10)


def test_adam_matches_torch_optim():
    L = _lib.lib()
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(5000, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(1, 6):
        grad = torch.randn(5000, generator=g) * (10.0 ** (t - 3))
        ref.grad = grad.clone()
        opt.step()
        gd = grad.cuda()
        _lib.check(L.egn_adam_step_f32(_lib.ptr(p), _lib.ptr(gd), _lib.ptr(m), _lib.ptr(v), 5000, 1e-3, 0.9, 0.999,
                                       1e-8, t, _st()))
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=0, atol=2e-6)


def test_running_stat_update_and_add():
    L = _lib.lib()
    r = torch.arange(10, dtype=torch.float32).cuda()
    b = torch.ones(10, device='cuda') * 2
    _lib.check(L.egn_ema_f32(_lib.ptr(r), _lib.ptr(b), 0.1, 10, _st()))
    np.testing.assert_allclose(r.cpu().numpy(), 0.9 * np.arange(10) + 0.2, rtol=1e-6)
    a = torch.randn(64, 36, device='cuda')
    c = torch.randn(64, 36, device='cuda')
    y = torch.zeros_like(a)
    _lib.check(L.egn_add_f32(_lib.ptr(a), _lib.ptr(c), _lib.ptr(y), a.numel(), _st()))
    assert torch.equal(y, a + c)


def _nhwc(t, cs):
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, cs)
    out[..., :c] = t.permute(0, 2, 3, 1)
    return out.contiguous()


@pytest.mark.parametrize('n,cin,cout,h,w,k,stride,pad', [
    (100, 66, 128, 1, 1, 1, 1, 0),       # Linear 66 -> 128 (ld 68), ragged batch
    (4096, 1024, 96, 1, 1, 1, 1, 0),     # lifter output layer at the bench batch
    (64, 128, 128, 1, 1, 1, 1, 0),       # 2x2-wave variant, one tile in each direction
    (3, 48, 48, 12, 20, 3, 1, 1),        # 3x3 with partial spatial tiles (Winograd form: 48-multiples, s1 p1)
    (2, 96, 192, 16, 16, 3, 1, 1),       # Winograd form, several co / ci tiles, one 16-wide tile per row
    (3, 48, 96, 8, 8, 3, 1, 1),          # Winograd form on 8x8 maps: image pairs, odd batch
    (2, 48, 48, 5, 7, 3, 1, 1),          # odd map sizes: the last Winograd tiles hang over the border
    (5, 192, 48, 4, 4, 3, 1, 1),
    (1, 48, 48, 40, 40, 3, 1, 1),        # more tiles than K splits: several stages per block
    (2, 96, 200, 16, 16, 3, 1, 1),       # several co / ci tiles
    (2, 48, 96, 16, 16, 3, 2, 1),        # strided 3x3 (fuse-layer down path)
    (5, 3, 64, 16, 16, 3, 2, 1),         # stem: 3 input channels (cs 4)
    (2, 96, 48, 8, 8, 1, 1, 0),          # 1x1 fuse conv
    (2, 35, 66, 8, 8, 1, 2, 0),          # 1x1 stride-2 projection (head2)
    (6, 66, 66, 4, 4, 4, 1, 0),          # 4x4 valid conv of the coordinate head
    (3, 10, 10, 4, 3, (4, 3), 1, 0),     # 4x3 valid conv (Pedestrian 64x48 maps)
])
def test_conv_wgrad(n, cin, cout, h, w, k, stride, pad):
    L = _lib.lib()
    g = torch.Generator().manual_seed(n * 131 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    kh, kw = k if isinstance(k, tuple) else (k, k)
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    dy = torch.randn(n, cout, ho, wo, generator=g)
    wt = torch.zeros(cout, cin, kh, kw, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double(), wt, None, stride, pad).backward(dy.double())
    cs_in, cs_out = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    xd, dyd = _nhwc(x, cs_in).cuda(), _nhwc(dy, cs_out).cuda()
    need = L.egn_conv2d_wgrad_ws_bytes(n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride, pad)
    assert need > 0
    ws = torch.zeros(need // 4, device='cuda')
    dw = torch.full((cout, cin, kh, kw), 3.0, device='cuda')
    _lib.check(L.egn_conv2d_wgrad_f32(_lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(dw), n, h, w, cin, cs_in, cout, cs_out,
                                      kh, kw, stride, pad, _lib.ptr(ws), need, _st()))
    want = wt.grad
    tol = 2e-6 * float(want.abs().max()) * max(1.0, (n * ho * wo) ** 0.5 / 8)
    np.testing.assert_allclose(dw.cpu().numpy(), want.numpy(), rtol=0, atol=tol)
    # deterministic: a second launch gives the same bits
    dw2 = torch.zeros_like(dw)
    _lib.check(L.egn_conv2d_wgrad_f32(_lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(dw2), n, h, w, cin, cs_in, cout, cs_out,
                                      kh, kw, stride, pad, _lib.ptr(ws), need, _st()))
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize('n,cin,cout,h,w,k,stride,pad', [
    (32, 48, 48, 64, 64, 3, 1, 1),       # BASELINE config 4 per-GPU sizes: the full-resolution branch,
    (32, 96, 192, 32, 32, 3, 2, 1),      # a strided fuse conv,
    (32, 384, 384, 8, 8, 3, 1, 1),       # the coarsest branch,
    (32, 384, 48, 8, 8, 1, 1, 0),        # a 1x1 fuse conv,
    (4096, 1024, 1024, 1, 1, 1, 1, 0),   # and the lifter's Linear at config 3's batch
])
def test_forward_dgrad_wgrad_are_adjoint_at_full_size(n, cin, cout, h, w, k, stride, pad):
    """Size-independent property at the BASELINE sizes (no CPU oracle needed): the three
    convolution kernels are adjoints of one another,
        <conv(x, w), dy> = <w, wgrad(x, dy)> = <x, dgrad(dy, w)>,
    inner products accumulated in float64."""
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(n + cin)
    cs_in, cs_out = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = torch.zeros(n, h, w, cs_in, device='cuda')
    x[..., :cin] = torch.randn(n, h, w, cin, device='cuda', generator=g)
    dy = torch.zeros(n, ho, wo, cs_out, device='cuda')
    dy[..., :cout] = torch.randn(n, ho, wo, cout, device='cuda', generator=g)
    wt = torch.randn(cout, cin, k, k, device='cuda', generator=g) * 0.05
    one = torch.ones(max(cin, cout) + 16, device='cuda')
    zero = torch.zeros(max(cin, cout) + 16, device='cuda')
    wp = torch.zeros(L.egn_packed_weight_floats(cout, cin, k, k, 0), device='cuda')
    _lib.check(L.egn_pack_conv_weight_f32(_lib.ptr(wt), cout, cin, k, k, 0, _lib.ptr(wp), _st()))
    y = torch.empty(n, ho, wo, cs_out, device='cuda')
    _lib.check(L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(one), _lib.ptr(zero), None, _lib.ptr(y),
                                n, h, w, cin, cs_in, cout, cs_out, k, k, stride, pad, 0, 0, 0, _st()))
    need = L.egn_conv2d_wgrad_ws_bytes(n, h, w, cin, cs_in, cout, cs_out, k, k, stride, pad)
    ws = torch.empty(need // 4, device='cuda')
    dw = torch.empty_like(wt)
    _lib.check(L.egn_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), n, h, w, cin, cs_in, cout, cs_out,
                                      k, k, stride, pad, _lib.ptr(ws), need, _st()))
    wq = torch.zeros(L.egn_packed_weight_floats(cout, cin, k, k, 1), device='cuda')
    _lib.check(L.egn_pack_conv_weight_f32(_lib.ptr(wt), cout, cin, k, k, 1, _lib.ptr(wq), _st()))
    src, sh, sw = dy, ho, wo
    if stride == 2:
        src, sh, sw = torch.empty(n, h, w, cs_out, device='cuda'), h, w
        _lib.check(L.egn_zero_insert2_f32(_lib.ptr(dy), _lib.ptr(src), n, ho, wo, h, w, cs_out, _st()))
    dx = torch.empty(n, h, w, cs_in, device='cuda')
    _lib.check(L.egn_conv2d_f32(_lib.ptr(src), _lib.ptr(wq), _lib.ptr(one), _lib.ptr(zero), None, _lib.ptr(dx),
                                n, sh, sw, cout, cs_out, cin, cs_in, k, k, 1, k - 1 - pad, 0, 0, 0, _st()))
    a = float((y.double() * dy.double()).sum())
    b = float((wt.double() * dw.double()).sum())
    c = float((x.double() * dx.double()).sum())
    scale = float((y.double() ** 2).sum().sqrt() * (dy.double() ** 2).sum().sqrt())
    assert abs(a - b) < 2e-6 * scale and abs(a - c) < 2e-6 * scale, (a, b, c, scale)


@pytest.mark.parametrize('n,cin,cout,h,w,k,stride,pad', [
    (2, 48, 48, 12, 20, 3, 1, 1),
    (2, 96, 200, 16, 16, 3, 1, 1),
    (2, 48, 96, 16, 16, 3, 2, 1),        # strided: zero-insert + stride-1 conv
    (3, 96, 48, 8, 8, 1, 1, 0),
    (2, 35, 66, 8, 8, 1, 2, 0),          # 1x1 stride 2
    (6, 66, 66, 4, 4, 4, 1, 0),          # 4x4 valid -> "full" correlation, pad 3
    (64, 1024, 96, 1, 1, 1, 1, 0),       # Linear
])
def test_conv_forward_and_dgrad_with_device_packed_weights(n, cin, cout, h, w, k, stride, pad):
    """egn_pack_conv_weight_f32 (forward and data-gradient filters) + egn_conv2d_f32
    (+ egn_zero_insert2_f32) against F.conv2d and its autograd input gradient."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(cin * 3 + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(n, cout, ho, wo, generator=g)
    xx = x.double().requires_grad_(True)
    yy = torch.nn.functional.conv2d(xx, wt.double(), None, stride, pad)
    yy.backward(dy.double())
    cs_in, cs_out = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    wd = wt.cuda()
    one = torch.ones(max(cin, cout) + 16, device='cuda')
    zero = torch.zeros(max(cin, cout) + 16, device='cuda')
    # forward
    wp = torch.zeros(L.egn_packed_weight_floats(cout, cin, k, k, 0), device='cuda')
    _lib.check(L.egn_pack_conv_weight_f32(_lib.ptr(wd), cout, cin, k, k, 0, _lib.ptr(wp), _st()))
    xd = _nhwc(x, cs_in).cuda()
    yd = torch.full((n, ho, wo, cs_out), 5.0, device='cuda')
    _lib.check(L.egn_conv2d_f32(_lib.ptr(xd), _lib.ptr(wp), _lib.ptr(one), _lib.ptr(zero), None, _lib.ptr(yd),
                                n, h, w, cin, cs_in, cout, cs_out, k, k, stride, pad, 0, 0, 0, _st()))
    got = yd.cpu()[..., :cout].permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), yy.detach().numpy(), rtol=0, atol=2e-5 * float(yy.abs().max()))
    assert float(yd.cpu()[..., cout:].abs().max() if cs_out > cout else 0.0) == 0.0
    # data gradient
    wq = torch.zeros(L.egn_packed_weight_floats(cout, cin, k, k, 1), device='cuda')
    _lib.check(L.egn_pack_conv_weight_f32(_lib.ptr(wd), cout, cin, k, k, 1, _lib.ptr(wq), _st()))
    dyd = _nhwc(dy, cs_out).cuda()
    if stride == 2:
        up = torch.full((n, h, w, cs_out), 9.0, device='cuda')
        _lib.check(L.egn_zero_insert2_f32(_lib.ptr(dyd), _lib.ptr(up), n, ho, wo, h, w, cs_out, _st()))
        src, sh, sw = up, h, w
    else:
        src, sh, sw = dyd, ho, wo
    dxd = torch.full((n, h, w, cs_in), 5.0, device='cuda')
    _lib.check(L.egn_conv2d_f32(_lib.ptr(src), _lib.ptr(wq), _lib.ptr(one), _lib.ptr(zero), None, _lib.ptr(dxd),
                                n, sh, sw, cout, cs_out, cin, cs_in, k, k, 1, k - 1 - pad, 0, 0, 0, _st()))
    got = dxd.cpu()[..., :cin].permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), xx.grad.numpy(), rtol=0, atol=2e-5 * float(xx.grad.abs().max()))


def test_fuse_backward_terms():
    """y = relu(t0 + up2(t1) + up4(t2)): per-term gradients = gated block sums."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(8)
    n, hh, ww, c = 2, 16, 8, 48
    ts = [torch.randn(n, c, hh >> s, ww >> s, generator=g, dtype=torch.float64, requires_grad=True) for s in (0, 1, 2)]
    y = torch.relu(ts[0] + torch.nn.functional.interpolate(ts[1], scale_factor=2, mode='nearest')
                   + torch.nn.functional.interpolate(ts[2], scale_factor=4, mode='nearest'))
    dy = torch.randn(n, c, hh, ww, generator=g)
    y.backward(dy.double())
    yd, dyd = _nhwc(y.detach().float(), c).cuda(), _nhwc(dy, c).cuda()
    for s in (0, 1, 2):
        out = torch.full((n, hh >> s, ww >> s, c), 4.0, device='cuda')
        _lib.check(L.egn_fuse_bwd_f32(_lib.ptr(dyd), _lib.ptr(yd), _lib.ptr(out), n, hh, ww, c, s, _st()))
        np.testing.assert_allclose(out.cpu().permute(0, 3, 1, 2).numpy(), ts[s].grad.numpy(), rtol=0, atol=1e-5)
    out = torch.zeros(n, hh >> 1, ww >> 1, c, device='cuda')
    _lib.check(L.egn_fuse_bwd_f32(_lib.ptr(dyd), None, _lib.ptr(out), n, hh, ww, c, 1, _st()))      # no gate
    want = torch.nn.functional.avg_pool2d(dy, 2) * 4
    np.testing.assert_allclose(out.cpu().permute(0, 3, 1, 2).numpy(), want.numpy(), rtol=0, atol=1e-5)


def test_sigmoid_backward_and_l1_loss():
    L = _lib.lib()
    g = torch.Generator().manual_seed(2)
    z = torch.randn(40, 66, generator=g, dtype=torch.float64, requires_grad=True)
    tgt = torch.rand(40, 66, generator=g)
    y = torch.sigmoid(z)
    loss = 0.1 * torch.nn.functional.l1_loss(y, tgt.double())
    loss.backward()
    yd, td = y.detach().float().cuda(), tgt.cuda()
    dyd = torch.zeros_like(yd)
    ld = torch.zeros(1, dtype=torch.float64, device='cuda')
    _lib.check(L.egn_l1_f32(_lib.ptr(yd), _lib.ptr(td), yd.numel(), 0.1, _lib.ptr(dyd), _lib.ptr(ld), _st()))
    assert abs(float(ld.item()) - float(loss.detach())) < 1e-7
    dz = torch.zeros_like(yd)
    _lib.check(L.egn_sigmoid_bwd_f32(_lib.ptr(dyd), _lib.ptr(yd), _lib.ptr(dz), yd.numel(), _st()))
    np.testing.assert_allclose(dz.cpu().numpy(), z.grad.numpy(), rtol=1e-5, atol=1e-9)


def test_conv_wgrad_refuses_unsupported():
    L = _lib.lib()
    a = torch.zeros(64, device='cuda')
    assert L.egn_conv2d_wgrad_ws_bytes(1, 8, 8, 4, 4, 4, 4, 5, 5, 1, 0) < 0           # 25 taps
    assert L.egn_conv2d_wgrad_f32(_lib.ptr(a), _lib.ptr(a), _lib.ptr(a), 1, 8, 8, 4, 6, 4, 4, 3, 3, 1, 1,
                                  _lib.ptr(a), 256, _st()) != 0                       # cs_in % 4


def test_all_filters_packed_in_one_launch_equal_the_single_packs():
    from egonet_amd.train_hrnet import PackedFilters
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    ws = [torch.randn(*s, generator=g).cuda() for s in ((48, 48, 3, 3), (33, 48, 1, 1), (66, 35, 3, 3), (66, 66, 4, 4))]
    pf = PackedFilters(torch.device('cuda', torch.cuda.current_device()))
    assert pf.pack_all(_st()) is False                        # nothing known yet
    for w in ws:
        for dgrad in (0, 1):
            pf.get(w, dgrad, _st())
    pf.finalize()
    for w in ws:                                              # the optimizer step changes the weights in place
        w.mul_(1.5).add_(0.25)
    assert pf.pack_all(_st()) is True
    for w in ws:
        cout, cin, kh, kw = w.shape
        for dgrad in (0, 1):
            want = torch.zeros(L.egn_packed_weight_floats(cout, cin, kh, kw, dgrad), device='cuda')
            _lib.check(L.egn_pack_conv_weight_f32(_lib.ptr(w), cout, cin, kh, kw, dgrad, _lib.ptr(want), _st()))
            assert torch.equal(pf.get(w, dgrad, _st()), want)
    ws[1].data = ws[1].data.clone()                           # storage moved: the table is rebuilt, not trusted
    assert pf.pack_all(_st()) is False


def test_f43_filters_in_the_one_launch_packer_equal_the_single_packs():
    """[round 5] descriptor code 4 (+ dgrad): the F(4x4,3x3) filter of conv_wino4.hip packed with all the others
    (egn_pack_conv_weights_batch_f32) == egn_wino4_pack_weight_f32 bit for bit (and engine.pack_wino4_weight, the host
    transform the inference engine uses, to the last fp32 bit or two), beside direct and F(2x2,3x3) entries of the same table."""
    from egonet_amd import engine
    from egonet_amd.train_hrnet import PackedFilters
    L = _lib.lib()
    g = torch.Generator().manual_seed(12)
    ws = [torch.randn(*s, generator=g).cuda() for s in ((48, 48, 3, 3), (96, 48, 3, 3), (48, 192, 3, 3), (384, 32, 3, 3))]
    pf = PackedFilters(torch.device('cuda', torch.cuda.current_device()))
    for w in ws:
        for dgrad in (0, 1):
            cout, cin = w.shape[:2]
            if L.egn_wino4_pack_weight_floats(cout, cin, dgrad) > 0:
                pf.get(w, dgrad, _st(), wino=3)
            pf.get(w, dgrad, _st(), wino=True)
            pf.get(w, dgrad, _st())
    pf.finalize()
    for w in ws:
        w.mul_(0.75).add_(0.125)
    for _, wp in pf.entries.values():
        wp.fill_(float('nan'))
    assert pf.pack_all(_st()) is True
    n43 = 0
    for w in ws:
        cout, cin = w.shape[:2]
        for dgrad in (0, 1):
            nfl = L.egn_wino4_pack_weight_floats(cout, cin, dgrad)
            if nfl <= 0:
                continue
            n43 += 1
            want = torch.full((nfl,), float('nan'), device='cuda')
            _lib.check(L.egn_wino4_pack_weight_f32(_lib.ptr(w), cout, cin, dgrad, _lib.ptr(want), _st()))
            got = pf.get(w, dgrad, _st(), wino=3)
            assert not torch.isnan(got).any() and torch.equal(got, want)
            if not dgrad:       # the host transform (einsum with G in float64) rounds the last bit differently here and there
                np.testing.assert_allclose(got.cpu().numpy(), engine.pack_wino4_weight(w.cpu()).reshape(-1).numpy(),
                                           rtol=3e-7, atol=1e-9)
            nf2 = L.egn_wino_weight_floats(cout, cin, dgrad)
            want2 = torch.zeros(nf2, device='cuda')
            _lib.check(L.egn_wino_pack_weight_f32(_lib.ptr(w), cout, cin, dgrad, _lib.ptr(want2), _st()))
            assert torch.equal(pf.get(w, dgrad, _st(), wino=True), want2)
    assert n43 >= 6


@pytest.mark.parametrize('n,h,w,cin,cout,cfg', [
    (5, 64, 64, 48, 48, 70),      # 16 x 32 regions, two m-tiles, several items per block
    (3, 32, 32, 96, 96, 70),      # two co-tiles
    (7, 16, 16, 192, 192, 80),    # 16 x 16 regions, 16-channel stages, four co-tiles (co-tiles on the XCDs)
    (6, 16, 16, 96, 48, 84),      # ... K split over two blocks (ticket words)
    (9, 8, 8, 384, 384, 83),      # four 8 x 8 images per region (the last one partial), K split, eight co-tiles
    (5, 8, 8, 64, 96, 82),        # ... without the split
    (700, 16, 16, 32, 48, 80),    # more items than blocks: a block's items share its co-tile
])
def test_bn_statistics_fused_into_the_f43_item_end(n, h, w, cin, cout, cfg):
    """[round 5] conv_wino4s_kernel through egn_conv2d_ex_f32 (partials, ticket words): the conv output bit for bit
    that of the inference build (egn_conv2d_f32, for the K split its three-launch form), and finalised statistics ==
    egn_bn_stats_f32 on that output; the ticket words are zero again afterwards."""
    from egonet_amd import engine
    L = _lib.lib()
    st = _lib.current_stream()
    g = torch.Generator().manual_seed(cfg + n)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    wu = engine.pack_wino4_weight(wt).reshape(-1).cuda()
    ones, zeros = torch.ones(cout + 16).cuda(), torch.zeros(cout + 16).cuda() + 0.5          # a shift: nonzero means
    rows = n * h * w
    dims = (n, h, w, cin, cin, cout, cout, 3, 3, 1, 1)
    nrows = L.egn_conv2d_bnstats_rows(*dims, cfg)
    ntk = L.egn_conv2d_ticket_words(*dims, cfg)
    assert nrows > 0 and (ntk > 0) == (cfg in (83, 84))
    tickets = torch.zeros(max(ntk, 1) + 3, dtype=torch.int32, device='cuda')
    part = torch.full((nrows * 2 * cout,), float('nan'), dtype=torch.float64, device='cuda')
    y1, y2 = torch.empty(n, h, w, cout).cuda(), torch.empty(n, h, w, cout).cuda()
    for _ in range(2):          # twice: the words are left zero
        _lib.check(L.egn_conv2d_ex_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(ones), _lib.ptr(zeros), None, _lib.ptr(y1),
                                       *dims, 0, cfg, _lib.ptr(part), nrows, _lib.ptr(tickets), tickets.numel(), st))
    _lib.check(L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(ones), _lib.ptr(zeros), None, _lib.ptr(y2),
                                *dims, 0, 0, cfg, st))
    assert torch.equal(y1, y2) and not torch.isnan(part).any() and int(tickets.abs().sum()) == 0
    if ntk > 0:                 # statistics need the one-kernel form; too few words are refused
        assert L.egn_conv2d_ex_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(ones), _lib.ptr(zeros), None, _lib.ptr(y1),
                                   *dims, 0, cfg, _lib.ptr(part), nrows, None, 0, st) != 0
        assert L.egn_conv2d_ex_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(ones), _lib.ptr(zeros), None, _lib.ptr(y1),
                                   *dims, 0, cfg, None, 0, _lib.ptr(tickets), ntk - 1, st) != 0
    assert L.egn_conv2d_ex_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(ones), _lib.ptr(zeros), None, _lib.ptr(y1),
                               *dims, 0, cfg, _lib.ptr(part), nrows - 1, _lib.ptr(tickets), tickets.numel(), st) != 0
    outs = []
    for fused in (True, False):
        mean, istd, varu = (torch.empty(cout).cuda() for _ in range(3))
        rm, rv = torch.zeros(cout).cuda() + 0.25, torch.ones(cout).cuda()
        if fused:
            _lib.check(L.egn_bn_stats_finalize_f32(_lib.ptr(part), nrows, rows, cout, 1e-5, _lib.ptr(mean),
                                                   _lib.ptr(istd), _lib.ptr(varu), _lib.ptr(rm), _lib.ptr(rv), 0.1, st))
        else:
            ws = torch.empty(L.egn_colreduce_ws_bytes(cout) // 4).cuda()
            _lib.check(L.egn_bn_stats_f32(_lib.ptr(y2), rows, cout, cout, 1e-5, _lib.ptr(mean), _lib.ptr(istd),
                                          _lib.ptr(varu), _lib.ptr(rm), _lib.ptr(rv), 0.1, _lib.ptr(ws), st))
        outs.append([t.cpu().numpy() for t in (mean, istd, varu, rm, rv)])
    z = y2.double().reshape(rows, cout).cpu()
    np.testing.assert_allclose(outs[0][0], z.mean(0).numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(outs[0][1], (1.0 / torch.sqrt(z.var(0, unbiased=False) + 1e-5)).numpy(), rtol=2e-5)
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_allclose(a, b, rtol=3e-6, atol=3e-7)


def test_bad_arguments_are_refused():
    L = _lib.lib()
    a = torch.zeros(16, device='cuda')
    assert L.egn_add_f32(_lib.ptr(a), _lib.ptr(a), _lib.ptr(a), 6, _st()) != 0           # n % 4
    assert L.egn_transpose_f32(_lib.ptr(a), 4, 4, 2, _lib.ptr(a), 4, _st()) != 0          # ld_src < C
    assert L.egn_adam_step_f32(_lib.ptr(a), _lib.ptr(a), _lib.ptr(a), _lib.ptr(a), 16, 1e-3, 0.9, 0.999, 1e-8, 0,
                               _st()) != 0                                                 # step counts from 1
    assert L.egn_colsum_f32(_lib.ptr(a), 4, 4, 4, _lib.ptr(a), None, _st()) != 0          # no workspace


@pytest.mark.parametrize('tag', ['s1', 's2', 'rect'])
def test_gaussian_targets_vs_reference(tag):
    """csrc/targets.hip against the reference's generate_target outputs
    (tests/golden/targets.npz): same float32 formula; expf vs numpy's exp may
    differ in the last bit."""
    from conftest import golden
    from egonet_amd.common import img_proc
    g = golden('targets.npz')
    prm = dict(num_joints=12, target_type='gaussian', input_size=g[tag + '/input_size'],
               heatmap_size=g[tag + '/heatmap_size'], sigma=int(g[tag + '/sigma']), use_different_joints_weight=False)
    t, w = img_proc.generate_target_batch(g[tag + '/joints'], g[tag + '/vis'], prm)
    want = g[tag + '/target']
    assert tuple(t.shape) == want.shape
    np.testing.assert_allclose(t.cpu().numpy(), want, rtol=0, atol=2e-7)
    assert np.array_equal(t.cpu().numpy() != 0, want != 0)              # support of every dot: exact
    np.testing.assert_array_equal(w.cpu().numpy(), g[tag + '/weight'])
    t1, w1 = img_proc.generate_target(g[tag + '/joints'][1], g[tag + '/vis'][1], prm)    # the per-sample API
    np.testing.assert_allclose(t1, want[1], rtol=0, atol=2e-7)
    np.testing.assert_array_equal(w1, g[tag + '/weight'][1])


CR_CASES = [(t, s, th) for t in ('rand', 'wide', 'cluster') for s in ('sl1', 'l1', 'mse') for th in (0.15, 0.1)]


def _cross_ratio(coords, idx, thres, spec, weight, dcoords=None, target_cr=4 / 3):
    L = _lib.lib()
    n, k = coords.shape[:2]
    idx_d = torch.as_tensor(idx, dtype=torch.int32).contiguous().cuda()
    nl = idx_d.shape[0]
    ws = torch.empty(L.egn_cross_ratio_ws_bytes(n, nl) // 4, device='cuda')
    loss = torch.zeros(1, dtype=torch.float64, device='cuda')
    _lib.check(L.egn_cross_ratio_f32(_lib.ptr(coords), n, k, _lib.ptr(idx_d), nl, target_cr, thres,
                                     {'mse': 0, 'l1': 1, 'sl1': 2}[spec], weight, _lib.ptr(dcoords), _lib.ptr(loss),
                                     _lib.ptr(ws), _st()), 'cross_ratio')
    return float(loss.item()), ws.view(n, nl, -1)


@pytest.mark.parametrize('tag,spec,thres', CR_CASES)
def test_cross_ratio_term_vs_reference(tag, spec, thres):
    """csrc/cross_ratio.hip against the REFERENCE's calc_cross_ratio_loss / get_cr_mask
    (tests/golden/cr_loss.npz): value, kept lines, gradient w.r.t. the coordinates."""
    from conftest import golden
    g = golden('cr_loss.npz')
    key = '%s/%s/%g' % (tag, spec, thres)
    c = torch.from_numpy(g[tag + '/coords']).cuda()
    dc = torch.zeros_like(c)
    loss, ws = _cross_ratio(c, g['cr_indices'], thres, spec, 1.0, dc)
    np.testing.assert_array_equal(ws[..., 9].cpu().numpy(), g[key + '/mask'][..., 0])      # bit exact line mask
    ref = g[key + '/grad']
    np.testing.assert_allclose(loss, float(g[key + '/loss']), rtol=5e-6, atol=0)
    np.testing.assert_allclose(dc.cpu().numpy(), ref, rtol=0, atol=5e-6 * max(1.0, float(np.abs(ref).max())))


def test_cross_ratio_term_accumulates_and_scales():
    from conftest import golden
    g = golden('cr_loss.npz')
    c = torch.from_numpy(g['rand/coords']).cuda()
    base = torch.full_like(c, 0.25)
    dc = base.clone()
    loss, _ = _cross_ratio(c, g['cr_indices'], 0.15, 'sl1', 0.05, dc)
    ref = g['rand/sl1/0.15/grad']
    np.testing.assert_allclose(loss, 0.05 * float(g['rand/sl1/0.15/loss']), rtol=5e-6)
    np.testing.assert_allclose((dc - base).cpu().numpy(), 0.05 * ref, rtol=0, atol=1e-6 * float(np.abs(ref).max()))
    loss2, _ = _cross_ratio(c, g['cr_indices'], 0.15, 'sl1', 0.05, None)          # value only
    assert loss2 == loss
    L = _lib.lib()
    assert L.egn_cross_ratio_f32(_lib.ptr(c), 6, 33, None, 12, 4 / 3, 0.15, 2, 1.0, None, None, None, _st()) != 0


@pytest.mark.parametrize('n,h,w,cin,cout,cfg', [
    (3, 32, 32, 48, 48, 51),      # 8 waves, 16 x 16 tiles
    (5, 8, 8, 32, 96, 52),        # four 8 x 8 images per block, partial batch, two co-tiles
    (3, 8, 8, 48, 48, 56),        # 4 waves, two images per block
    (2, 24, 16, 16, 144, 57),     # 4 waves, 8 x 16 tiles, three co-tiles
])
def test_bn_statistics_fused_into_the_winograd_conv_epilogue(n, h, w, cin, cout, cfg):
    """egn_conv2d_bnstats_f32 + egn_bn_stats_finalize_f32 == egn_conv2d_f32 + egn_bn_stats_f32: same
    conv output bit for bit, mean / invstd / running statistics to fp32 rounding (the partial sums are
    grouped per (tile, wave) instead of per row block)."""
    from egonet_amd import _lib, engine
    L = _lib.lib()
    st = _lib.current_stream()
    g = torch.Generator().manual_seed(cfg)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    wu = engine.pack_wino_weight(wt).cuda()
    ones, zeros = torch.ones(cout + 16).cuda(), torch.zeros(cout + 16).cuda()
    rows = n * h * w
    nrows = L.egn_conv2d_bnstats_rows(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, cfg)
    assert nrows > 0 and L.egn_conv2d_bnstats_rows(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 16) == 0
    part = torch.full((nrows * 2 * cout,), float('nan'), dtype=torch.float64, device='cuda')
    y1, y2 = torch.empty(n, h, w, cout).cuda(), torch.empty(n, h, w, cout).cuda()
    _lib.check(L.egn_conv2d_bnstats_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(ones), _lib.ptr(zeros), _lib.ptr(y1),
                                        n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, cfg, _lib.ptr(part), nrows, st))
    _lib.check(L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(ones), _lib.ptr(zeros), None, _lib.ptr(y2),
                                n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 0, cfg, st))
    assert torch.equal(y1, y2) and not torch.isnan(part).any()
    outs = []
    for fused in (True, False):
        mean, istd, varu = (torch.empty(cout).cuda() for _ in range(3))
        rm, rv = torch.zeros(cout).cuda() + 0.25, torch.ones(cout).cuda()
        if fused:
            _lib.check(L.egn_bn_stats_finalize_f32(_lib.ptr(part), nrows, rows, cout, 1e-5, _lib.ptr(mean),
                                                   _lib.ptr(istd), _lib.ptr(varu), _lib.ptr(rm), _lib.ptr(rv), 0.1, st))
        else:
            ws = torch.empty(L.egn_colreduce_ws_bytes(cout) // 4).cuda()
            _lib.check(L.egn_bn_stats_f32(_lib.ptr(y2), rows, cout, cout, 1e-5, _lib.ptr(mean), _lib.ptr(istd),
                                          _lib.ptr(varu), _lib.ptr(rm), _lib.ptr(rv), 0.1, _lib.ptr(ws), st))
        outs.append([t.cpu().numpy() for t in (mean, istd, varu, rm, rv)])
    z = y2.double().reshape(rows, cout).cpu()
    np.testing.assert_allclose(outs[0][0], z.mean(0).numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(outs[0][1], (1.0 / torch.sqrt(z.var(0, unbiased=False) + 1e-5)).numpy(), rtol=2e-5)
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_allclose(a, b, rtol=3e-6, atol=3e-7)
