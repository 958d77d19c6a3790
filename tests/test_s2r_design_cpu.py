"""csrc/conv_s2r.hip without a GPU: the kernel's index algebra restated in numpy on the product code's filter packing
(engine.pack_conv_weight) -- item -> (tile, co-group), the im2col gather of the LDS-DMA (source offsets incl. zero padding,
lane-linear destination, quad swizzle), the MFMA fragment reads and their banks, the filter registers of a lane, the
epilogue addresses -- against torch's stride-2 convolution."""
import numpy as np
import torch
import torch.nn.functional as F

from egonet_amd import engine

OOB = 0xF0000000
CIN, CG, TH, TW = 48, 48, 2, 8


def f_swz(row):
    return ((row >> 3) & 1) << 1


def emulate(x, w, scale, shift, relu):
    """x [N,H,W,48] NHWC, w torch [Cout,48,3,3] -> y [N,Ho,Wo,Cout], every lane's arithmetic as the kernel does it."""
    N, H, W, _ = x.shape
    Cout = w.shape[0]
    CoP = (Cout + 15) // 16 * 16
    Ho, Wo = H // 2, W // 2
    wp = engine.pack_conv_weight(w).numpy().reshape(-1)            # [chunk][tap][quad][CoutP][4]
    xf = x.reshape(-1)
    y = np.full((N, Ho, Wo, Cout), np.nan, dtype=np.float32)
    ncg = Cout // CG
    tiles_x, tiles_y = Wo // TW, Ho // TH
    nitem = N * tiles_y * tiles_x * ncg
    grid = 8 * ncg                                                   # any multiple of ncg
    written = np.zeros((N, Ho, Wo, Cout), dtype=int)
    for blk in range(grid):
        cg = blk % ncg
        for item in range(blk, nitem, grid):
            assert item % ncg == cg                                  # the block's co-group is fixed
            t = item // ncg
            tx, tyn = t % tiles_x, t // tiles_x
            ty, n = tyn % tiles_y, tyn // tiles_y
            lds = np.full(27 * 256, np.nan, dtype=np.float32)        # 27 KB in floats
            for wave in range(3):
                for j in range(9):
                    p = 9 * wave + j
                    tap, ch = p // 3, p % 3
                    ky, kx = tap // 3, tap % 3
                    for lane in range(64):
                        drow = lane >> 2
                        dquad = (lane & 3) ^ f_swz(drow)
                        oy, ox = TH * ty + (drow >> 3), TW * tx + (drow & 7)
                        iy0, ix0 = 2 * oy - 1, 2 * ox - 1
                        pixoff = (((n * H + iy0) * W + ix0) * CIN + 4 * dquad) * 4
                        ok = 0 <= iy0 + ky < H and 0 <= ix0 + kx < W
                        voff = pixoff + ((ky * W + kx) * CIN + 16 * ch) * 4 if ok else OOB
                        dst = (p * 1024 + lane * 16) // 4
                        if voff >= xf.size * 4:
                            lds[dst:dst + 4] = 0.0
                        else:
                            assert voff >= 0 and voff % 16 == 0
                            lds[dst:dst + 4] = xf[voff // 4: voff // 4 + 4]
            assert not np.isnan(lds).any()
            # the MFMA itself: D[row][col] += sum_k A[row][k] B[k][col]; lane (li, kq) supplies A[li][kq], B[kq][li]
            acc = np.zeros((3, 16, 16), dtype=np.float64)            # [wave][row][col]
            for wave in range(3):
                for tap in range(9):
                    for ch in range(3):
                        for s in range(4):
                            A = np.zeros((16, 4))
                            B = np.zeros((4, 16))
                            for lane in range(64):
                                li, kq = lane & 15, lane >> 4
                                frag = li * 64 + ((kq ^ f_swz(li)) << 4)
                                A[li, kq] = lds[((tap * 3 + ch) * 1024 + frag) // 4 + s]
                                co = cg * CG + 16 * wave + li
                                B[kq, li] = wp[(((ch * 9 + tap) * 4 + kq) * CoP + co) * 4 + s]
                            acc[wave] += A @ B
            for wave in range(3):
                for lane in range(64):
                    li, kq = lane & 15, lane >> 4
                    co = cg * CG + 16 * wave + li
                    vo = (((n * Ho + TH * ty + (kq >> 1)) * Wo + TW * tx + 4 * (kq & 1)) * Cout + co) * 4
                    for r in range(4):
                        v = acc[wave][4 * kq + r, li] * scale[co] + shift[co]
                        if relu:
                            v = max(v, 0.0)
                        off = (vo + r * Cout * 4) // 4
                        idx = np.unravel_index(off, y.shape)
                        y[idx] = v
                        written[idx] += 1
    assert (written == 1).all()                                      # every output written exactly once
    return y


def test_s2r_emulation_equals_the_stride2_convolution():
    g = torch.Generator().manual_seed(5)
    for (N, H, W, Cout, relu) in ((2, 8, 16, 48, True), (1, 4, 32, 96, False)):
        x = torch.randn(N, H, W, CIN, generator=g)
        w = torch.randn(Cout, CIN, 3, 3, generator=g) / 20
        scale = (0.5 + torch.rand(Cout, generator=g)).numpy()
        shift = torch.randn(Cout, generator=g).numpy()
        got = emulate(x.numpy(), w, scale, shift, relu)
        want = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 2, 1)
        want = want * torch.from_numpy(scale).double()[None, :, None, None] + torch.from_numpy(shift).double()[None, :, None, None]
        if relu:
            want = F.relu(want)
        want = want.permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


def test_s2r_fragment_reads_are_bank_conflict_free():
    """ds_read_b128: four groups of 16 lanes (MI355X_MICROARCH.md), bank = (address / 4) mod 64."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for g in groups:
        banks = set()
        for lane in g:
            li, kq = lane & 15, lane >> 4
            frag = li * 64 + ((kq ^ f_swz(li)) << 4)
            for d in range(4):
                bk = (frag // 4 + d) % 64
                assert bk not in banks, (lane, bk)
                banks.add(bk)
