"""csrc/gemm.hip (the lifter's dense layers on the fp32 matrix pipe, reference libs/model/FCmodel.py:33-43,
92-105) through the C ABI against float64 products, and the in-kernel Philox dropout of the BatchNorm kernels
against a numpy restatement of Philox4x32-10."""
import numpy as np
import pytest
import torch

from egonet_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('form,M,N,K,variants', [
    (0, 256, 128, 64, (0, 1, 2, 3)),       # NT: z = a W^T + b
    (0, 4096, 1024, 1024, (3,)),           # the lifter's forward GEMM at the bench batch
    (1, 256, 256, 96, (0, 1, 2)),          # NN: da = dz W
    (1, 4096, 1024, 1024, (0,)),
    (2, 1024, 1024, 256, (0, 1)),          # TN: dW = dz^T a (64 tiles, no split)
    (2, 1024, 1024, 4096, (1,)),           # ... split along the batch, fixed-order reduction
])
def test_gemm_forms_vs_float64(form, M, N, K, variants):
    L = _lib.lib()
    st = _lib.current_stream()
    g = torch.Generator().manual_seed(form * 7 + K)
    if form == 0:
        A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
        want = A.double() @ B.double().t()
    elif form == 1:
        A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) / K ** 0.5
        want = A.double() @ B.double()
    else:
        A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g) / K ** 0.5
        want = A.double().t() @ B.double()
    bias = torch.randn(N, generator=g) if form == 0 else None
    if bias is not None:
        want = want + bias.double()
    assert L.egn_gemm_supported(form, M, N, K, A.shape[1], B.shape[1], N) == 1
    Ad, Bd = A.cuda(), B.cuda()
    bd = bias.cuda() if bias is not None else None
    need = L.egn_gemm_ws_bytes(form, M, N, K)
    ws = torch.empty(max(need // 4, 4), device='cuda')
    for v in variants:
        C = torch.full((M, N), float('nan'), device='cuda')
        _lib.check(L.egn_gemm_f32(form, _lib.ptr(Ad), _lib.ptr(Bd), _lib.ptr(C), _lib.ptr(bd), M, N, K, A.shape[1],
                                  B.shape[1], N, v, _lib.ptr(ws), need, st), 'gemm')
        torch.cuda.synchronize()
        err = float((C.double().cpu() - want).abs().max())
        assert err < 2e-6 * K ** 0.5 * float(want.abs().max()) + 1e-5, (form, v, err)
        C2 = torch.empty_like(C)
        _lib.check(L.egn_gemm_f32(form, _lib.ptr(Ad), _lib.ptr(Bd), _lib.ptr(C2), _lib.ptr(bd), M, N, K, A.shape[1],
                                  B.shape[1], N, v, _lib.ptr(ws), need, st), 'gemm')
        assert torch.equal(C, C2)                       # deterministic (fixed-order split-K reduction)
    # shapes the kernels do not take are refused (the callers keep the conv-kernel route for them)
    assert L.egn_gemm_supported(form, M + 1, N, K, A.shape[1], B.shape[1], N) == 0
    assert L.egn_gemm_supported(form, M, N, K + 4, A.shape[1] + 4, B.shape[1] + 4, N) == 0


def _philox4(k0, k1, c):
    """numpy Philox4x32-10 (Salmon et al.): c = uint32 [n, 4] counters -> [n, 4] draws."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c = c.astype(np.uint64)
    k0, k1 = int(k0), int(k1)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & np.uint64(0xffffffff), p1 >> np.uint64(32), p1 & np.uint64(0xffffffff)
        c = np.stack([hi1 ^ c[:, 1] ^ np.uint64(k0), lo1, hi0 ^ c[:, 3] ^ np.uint64(k1), lo0], axis=1)
        k0, k1 = (k0 + W0) & 0xffffffff, (k1 + W1) & 0xffffffff
    return c.astype(np.uint32)


def test_in_kernel_dropout_is_philox_and_forward_backward_agree():
    L = _lib.lib()
    st = _lib.current_stream()
    rows, cols, p, seed, layer = 37, 64, 0.5, 0x123456789ABCDEF, 3
    step = torch.tensor([11], dtype=torch.int32, device='cuda')
    mask = torch.empty(rows, cols, device='cuda')
    _lib.check(L.egn_dropout_mask_f32(_lib.ptr(mask), rows * cols, p, seed, _lib.ptr(step), layer, st))
    n4 = rows * cols // 4
    ctr = np.zeros((n4, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(n4)
    ctr[:, 2], ctr[:, 3] = layer, 11
    draws = _philox4(seed & 0xffffffff, seed >> 32, ctr).reshape(rows, cols)
    want = (draws >= np.uint32(int(p * 2 ** 32))).astype(np.float32)
    np.testing.assert_array_equal(mask.cpu().numpy(), want)
    assert 0.4 < want.mean() < 0.6
    # the *_drop_* kernels == the explicit-mask kernels fed with that mask, bit for bit
    g = torch.Generator().manual_seed(1)
    z, dy = torch.randn(rows, cols, generator=g).cuda(), torch.randn(rows, cols, generator=g).cuda()
    mean, istd = z.mean(0), (z.var(0, unbiased=False) + 1e-5).rsqrt()
    gm, bt = (torch.rand(cols, generator=g) + 0.5).cuda(), torch.randn(cols, generator=g).cuda()
    keep = 1.0 / (1.0 - p)
    ya, yb = torch.empty_like(z), torch.empty_like(z)
    _lib.check(L.egn_bn_act_fwd_f32(_lib.ptr(z), _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(gm), _lib.ptr(bt), _lib.ptr(mask),
                                    keep, 1, None, _lib.ptr(ya), rows, cols, cols, st))
    _lib.check(L.egn_bn_act_fwd_drop_f32(_lib.ptr(z), _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(gm), _lib.ptr(bt), p, seed,
                                         _lib.ptr(step), layer, 1, None, _lib.ptr(yb), rows, cols, cols, st))
    assert torch.equal(ya, yb) and float((ya == 0).float().mean()) > 0.5
    ws = torch.zeros(L.egn_colreduce_ws_bytes(cols) // 4, device='cuda')
    outs = []
    for drop in (False, True):
        db, dg, dz = torch.empty(cols, device='cuda'), torch.empty(cols, device='cuda'), torch.empty_like(z)
        if drop:
            _lib.check(L.egn_bn_bwd_sums_drop_f32(_lib.ptr(dy), _lib.ptr(z), p, seed, _lib.ptr(step), layer, _lib.ptr(mean),
                                                  _lib.ptr(istd), _lib.ptr(gm), _lib.ptr(bt), 1, None, rows, cols, cols,
                                                  _lib.ptr(db), _lib.ptr(dg), _lib.ptr(ws), st))
            _lib.check(L.egn_bn_bwd_dz_drop_f32(_lib.ptr(dy), _lib.ptr(z), p, seed, _lib.ptr(step), layer, _lib.ptr(mean),
                                                _lib.ptr(istd), _lib.ptr(gm), _lib.ptr(bt), 1, None, _lib.ptr(db), _lib.ptr(dg),
                                                _lib.ptr(dz), None, rows, cols, cols, st))
        else:
            _lib.check(L.egn_bn_bwd_sums_f32(_lib.ptr(dy), _lib.ptr(z), _lib.ptr(mask), keep, _lib.ptr(mean), _lib.ptr(istd),
                                             _lib.ptr(gm), _lib.ptr(bt), 1, None, rows, cols, cols, _lib.ptr(db), _lib.ptr(dg),
                                             _lib.ptr(ws), st))
            _lib.check(L.egn_bn_bwd_dz_f32(_lib.ptr(dy), _lib.ptr(z), _lib.ptr(mask), keep, _lib.ptr(mean), _lib.ptr(istd),
                                           _lib.ptr(gm), _lib.ptr(bt), 1, None, _lib.ptr(db), _lib.ptr(dg), _lib.ptr(dz), None,
                                           rows, cols, cols, st))
        outs.append((db, dg, dz))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # another step value -> another mask
    step.fill_(12)
    m2 = torch.empty_like(mask)
    _lib.check(L.egn_dropout_mask_f32(_lib.ptr(m2), rows * cols, p, seed, _lib.ptr(step), layer, st))
    assert not torch.equal(mask, m2)


@pytest.mark.parametrize('M,N,K,variant', [(256, 128, 64, 0), (4096, 1024, 1024, 3), (512, 256, 96, 1), (384, 128, 32, 2)])
def test_gemm_forward_epilogue_leaves_batchnorm_partial_sums(M, N, K, variant):
    """egn_gemm_ex_f32(form 0, stats): z = a W^T + b AND, from the same epilogue, partial column sums / sums of squares
    of z per 128-row block tile ([M / 128][2][N] doubles) -- nn.BatchNorm1d's batch statistics (FCmodel.py:33-43 in
    train mode) without a pass over z.  z is bit-identical to the plain call; the finalised mean / 1/sigma equal
    egn_bn_stats_f32's on the same z."""
    L = _lib.lib()
    st = _lib.current_stream()
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    z0 = torch.empty(M, N, device='cuda')
    _lib.check(L.egn_gemm_f32(0, _lib.ptr(A), _lib.ptr(B), _lib.ptr(z0), _lib.ptr(bias), M, N, K, K, K, N, variant, None, 0, st))
    nrow = L.egn_gemm_stats_rows(M)
    assert nrow == M // 128
    part = torch.full((nrow, 2, N), float('nan'), dtype=torch.float64, device='cuda')
    z1 = torch.empty(M, N, device='cuda')
    _lib.check(L.egn_gemm_ex_f32(0, _lib.ptr(A), _lib.ptr(B), _lib.ptr(z1), _lib.ptr(bias), None, _lib.ptr(part), nrow,
                                 M, N, K, K, K, N, variant, None, 0, st), 'gemm + stats')
    torch.cuda.synchronize()
    assert torch.equal(z0, z1)
    zd = z1.double().view(nrow, 128, N)
    np.testing.assert_allclose(part[:, 0].cpu().numpy(), zd.sum(1).cpu().numpy(), rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(part[:, 1].cpu().numpy(), (zd * zd).sum(1).cpu().numpy(), rtol=2e-6, atol=1e-4)
    # finalise == the two-pass statistics of the same z
    mean, istd, varu = (torch.empty(N, device='cuda') for _ in range(3))
    _lib.check(L.egn_bn_stats_finalize_f32(_lib.ptr(part), nrow, M, N, 1e-5, _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(varu),
                                           None, None, 0.1, st))
    m2, i2, v2 = (torch.empty(N, device='cuda') for _ in range(3))
    ws = torch.empty(L.egn_colreduce_ws_bytes(N) // 4, device='cuda')
    _lib.check(L.egn_bn_stats_f32(_lib.ptr(z1), M, N, N, 1e-5, _lib.ptr(m2), _lib.ptr(i2), _lib.ptr(v2), None, None, 0.1,
                                  _lib.ptr(ws), st))
    torch.cuda.synchronize()
    np.testing.assert_allclose(mean.cpu().numpy(), m2.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(istd.cpu().numpy(), i2.cpu().numpy(), rtol=2e-5, atol=0)
    # deterministic, and refused where it does not apply
    part2 = torch.empty_like(part)
    _lib.check(L.egn_gemm_ex_f32(0, _lib.ptr(A), _lib.ptr(B), _lib.ptr(z1), _lib.ptr(bias), None, _lib.ptr(part2), nrow,
                                 M, N, K, K, K, N, variant, None, 0, st))
    assert torch.equal(part, part2)
    assert L.egn_gemm_ex_f32(0, _lib.ptr(A), _lib.ptr(B), _lib.ptr(z1), None, None, _lib.ptr(part), nrow - 1, M, N, K, K, K,
                             N, variant, None, 0, st) != 0


def test_gemm_epilogue_statistics_with_a_large_column_mean():
    """[round 5, ADVICE r4] columns whose mean is ~1e3 x their deviation (a large bias): the epilogue accumulates sums
    and squares in doubles from the first add, so variance / 1/sigma of the finalise stage still agree with the
    float64 statistics of the stored z (fp32 squares lost several digits of E[x^2] - mean^2 there)."""
    L = _lib.lib()
    st = _lib.current_stream()
    M, N, K = 1024, 128, 64
    g = torch.Generator().manual_seed(77)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = (1000.0 + 50.0 * torch.randn(N, generator=g)).cuda()
    nrow = L.egn_gemm_stats_rows(M)
    part = torch.empty((nrow, 2, N), dtype=torch.float64, device='cuda')
    z = torch.empty(M, N, device='cuda')
    _lib.check(L.egn_gemm_ex_f32(0, _lib.ptr(A), _lib.ptr(B), _lib.ptr(z), _lib.ptr(bias), None, _lib.ptr(part), nrow,
                                 M, N, K, K, K, N, 0, None, 0, st))
    mean, istd, varu = (torch.empty(N, device='cuda') for _ in range(3))
    _lib.check(L.egn_bn_stats_finalize_f32(_lib.ptr(part), nrow, M, N, 1e-5, _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(varu),
                                           None, None, 0.1, st))
    torch.cuda.synchronize()
    zd = z.double()
    np.testing.assert_allclose(mean.cpu().numpy(), zd.mean(0).cpu().numpy(), rtol=2e-7)
    np.testing.assert_allclose(istd.cpu().numpy(), (zd.var(0, unbiased=False) + 1e-5).rsqrt().cpu().numpy(), rtol=1e-5)


def test_gemm_data_gradient_epilogue_adds_the_skip_path():
    """egn_gemm_ex_f32(form 1, addend): da = dz W + d_skip in one launch (the residual block's `out = x + y`,
    FCmodel.py:49-51, in the backward); equal to the plain product plus the addend, refused for the other forms."""
    L = _lib.lib()
    st = _lib.current_stream()
    M, N, K = 512, 256, 128
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g).cuda()
    B = torch.randn(K, N, generator=g).cuda()
    add = torch.randn(M, N, generator=g).cuda()
    c0, c1 = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    _lib.check(L.egn_gemm_f32(1, _lib.ptr(A), _lib.ptr(B), _lib.ptr(c0), None, M, N, K, K, N, N, 0, None, 0, st))
    _lib.check(L.egn_gemm_ex_f32(1, _lib.ptr(A), _lib.ptr(B), _lib.ptr(c1), None, _lib.ptr(add), None, 0, M, N, K, K, N, N, 0,
                                 None, 0, st))
    torch.cuda.synchronize()
    assert torch.equal(c1, c0 + add)
    assert L.egn_gemm_ex_f32(0, _lib.ptr(A), _lib.ptr(B), _lib.ptr(c1), None, _lib.ptr(add), None, 0, M, N, K, K, K, N, 0,
                             None, 0, st) != 0
