"""CPU emulation of the DESIGN of conv_wino43_kernel (csrc/conv_wino.hip; no GPU): halo slot map with skewed rows,
per-wave B^T row (which patch rows, which coefficients), the 6-point row transform, the packed F(4x4,3x3) filter
layout of engine.pack_wino43_weight, the v_mfma_f32_16x16x4_f32 operand / result mapping with k = 4 channels,
the output transform split (t_i per wave, A^T combination by waves 0..3) and the output addressing -- in numpy,
lane by lane, against a float64 direct convolution.  Also: the 2-way bank-conflict bound of the patch reads."""
import numpy as np
import torch

from egonet_amd import engine

RP, HS = 24, 448
ROWS = {0: ((0, 2, 2, 4), (4., -5., 0.)), 1: ((1, 2, 3, 4), (-4., -4., 1.)), 2: ((1, 2, 3, 4), (4., -4., -1.)),
        3: ((1, 2, 3, 4), (-2., -1., 2.)), 4: ((1, 2, 3, 4), (2., -1., -2.)), 5: ((1, 3, 3, 5), (4., -5., 0.))}
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def emulate(x, w, n_img, H, W, C, Co):
    """x [N,H,W,C], w [Co,C,3,3] -> y [N,H,W,Co] through the kernel's index maps (float64 arithmetic)."""
    U = engine.pack_wino43_weight(torch.from_numpy(w)).numpy().astype(np.float64)
    nct, nsteps = Co // 48, C // 4
    y = np.full((n_img, H, W, Co), np.nan)
    worst = 0
    for n in range(n_img):
        for ty in range(H // 16):
            for tx in range(W // 16):
                for ct in range(nct):
                    acc = np.zeros((6, 6, 3, 64, 4))          # [wave i][j][nt][lane][r]
                    for st in range(nsteps):
                        # --- halo stage image: slot e <- pixel (y, x) of the 18 x 18 patch, channels 4 st .. 4 st + 3
                        halo = np.full((HS, 4), np.nan)
                        for e in range(HS):
                            yy = e // RP
                            xx = e - yy * RP - (yy >> 2)
                            if yy < 18 and 0 <= xx < 18:
                                iy, ix = ty * 16 - 1 + yy, tx * 16 - 1 + xx
                                halo[e] = x[n, iy, ix, 4 * st:4 * st + 4] if (0 <= iy < H and 0 <= ix < W) else 0.0
                        slab = U[((ct * nsteps + st) * 36 * 4 * 48):((ct * nsteps + st + 1) * 36 * 4 * 48)].reshape(36, 4, 48)
                        for i in range(6):
                            (r0, r1, r2, r3), (c0, c1, c2) = ROWS[i]
                            V = np.zeros((6, 64))
                            banks = {}
                            for lane in range(64):
                                li, kq = lane & 15, lane >> 4
                                tyl, txl = li >> 2, li & 3

                                def rd(a, cc):
                                    slot = (4 * tyl + a) * RP + 4 * txl + tyl + (a >> 2) + cc
                                    if a == r0 and cc == 0:
                                        banks.setdefault(lane >> 5, []).append((slot * 4 + kq) % 32)
                                    v = halo[slot, kq]
                                    assert not np.isnan(v)
                                    return v
                                t = [c0 * rd(r0, cc) + c1 * rd(r1, cc) + c2 * rd(r2, cc) + rd(r3, cc) for cc in range(6)]
                                u_, v_ = t[4] - 4 * t[2], t[3] - 4 * t[1]
                                p_, q_ = t[4] - t[2], t[3] - t[1]
                                V[:, lane] = [4 * t[0] - 5 * t[2] + t[4], u_ + v_, u_ - v_, p_ + 2 * q_, p_ - 2 * q_,
                                              4 * t[1] - 5 * t[3] + t[5]]
                            for half in banks.values():
                                worst = max(worst, max(half.count(b) for b in set(half)))
                            for j in range(6):
                                for nt in range(3):
                                    A = np.zeros((16, 4)); B = np.zeros((4, 16))
                                    for lane in range(64):
                                        A[lane & 15, lane >> 4] = V[j, lane]
                                        B[lane >> 4, lane & 15] = slab[i * 6 + j, lane >> 4, nt * 16 + (lane & 15)]
                                    D = A @ B
                                    for lane in range(64):
                                        for r in range(4):
                                            acc[i, j, nt, lane, r] += D[4 * (lane >> 4) + r, lane & 15]
                    # --- output transform: t_i[b], exchange, waves 0..3 finish row a
                    m = acc
                    p_, q_, r_, s_ = m[:, 1] + m[:, 2], m[:, 1] - m[:, 2], m[:, 3] + m[:, 4], m[:, 3] - m[:, 4]
                    tb = np.stack([m[:, 0] + p_ + r_, q_ + 2 * s_, p_ + 4 * r_, q_ + 8 * s_ + m[:, 5]], axis=1)   # [i][b][nt][lane][r]
                    for a in range(4):
                        yv = np.einsum('i,ibnlr->bnlr', AT[a], tb)
                        for lane in range(64):
                            li, kq = lane & 15, lane >> 4
                            for nt in range(3):
                                for b in range(4):
                                    for r in range(4):
                                        oy, ox = ty * 16 + 4 * kq + a, tx * 16 + 4 * r + b
                                        co = ct * 48 + nt * 16 + li
                                        assert np.isnan(y[n, oy, ox, co])
                                        y[n, oy, ox, co] = yv[b, nt, lane, r]
    return y, worst


def test_wino43_design_matches_direct_convolution():
    rng = np.random.default_rng(0)
    N, H, W, C, Co = 1, 32, 16, 8, 48
    x = rng.standard_normal((N, H, W, C))
    w = rng.standard_normal((Co, C, 3, 3)) * 0.2
    y, worst = emulate(x, w, N, H, W, C, Co)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), padding=1)
    ref = ref.permute(0, 2, 3, 1).numpy()
    assert not np.isnan(y).any()
    # the packed filter is rounded to fp32 once: ~1e-7 relative
    np.testing.assert_allclose(y, ref, rtol=0, atol=5e-6 * np.abs(ref).max())
    assert worst <= 2, worst
