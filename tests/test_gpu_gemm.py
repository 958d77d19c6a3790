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
