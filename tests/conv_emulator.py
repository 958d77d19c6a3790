"""Lane-level numpy emulation of csrc/conv_mfma.hip (TEST INFRASTRUCTURE).

Re-states, formula for formula, what one workgroup of ``conv_mfma_kernel``
does: the halo-pixel offset table, the XOR-swizzled LDS images sA / sB filled
from NHWC input and the *packed* weights, the per-lane A/B fragment reads, the
v_mfma_f32_16x16x4_f32 operand/result lane mapping (A[i=l&15][k=l>>4],
B[k=l>>4][j=l&15], C[4*(l>>4)+r][l&15]) and the epilogue addressing.  It lets
the CPU test-suite validate the kernel *design* (weight packing done by the
product code, tile planning done by the library's host planner, fragment
mapping) against torch's conv2d without a GPU.  It is not a performance model
and not a fallback: nothing in egonet_amd/ imports it.
"""
import numpy as np

CK, CKQ = 16, 4


def mfma_16x16x4(a_lane, b_lane, c_tile):
    """a_lane, b_lane: [64] operand held by each lane; c_tile [16,16] += A @ B."""
    A = a_lane.reshape(4, 16).T          # A[i][k] from lane k*16 + i
    B = b_lane.reshape(4, 16)            # B[k][j] from lane k*16 + j
    # k-ordered fmaf chain like the hardware
    for k in range(4):
        c_tile += np.outer(A[:, k], B[k, :]).astype(np.float32)
    return c_tile


def emulate(x, wpack, scale, shift, res, plan, N, H, W, Cin, cs_in, Cout, cs_out, KH, KW, stride, pad,
            act, out_nchw):
    cfg, WM, WN, MT, NT, TH, TW, TNB, tps, lds, gx, gy = plan
    TN = WN * NT * 16
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    CoutP = (Cout + 15) // 16 * 16
    nchunk = (Cin + CK - 1) // CK
    taps = KH * KW
    HH, HW = (TH - 1) * stride + KH, (TW - 1) * stride + KW
    npix = TNB * HH * HW
    npixp = (npix + 15) // 16 * 16
    tiles_x, tiles_y = -(-Wo // TW), -(-Ho // TH)
    tile_px = TH * TW
    xf = x.reshape(-1)
    w4 = wpack.reshape(-1, 4)
    y = np.full((N, Cout, Ho, Wo) if out_nchw else (N, Ho, Wo, cs_out), np.nan, dtype=np.float32)
    lanes = np.arange(64)
    li, kq = lanes & 15, lanes >> 4
    act_id, res_after = act & 0xf, bool(act & 0x10)

    for tile in range(gx):
        tx, ty, tb = tile % tiles_x, (tile // tiles_x) % tiles_y, tile // (tiles_x * tiles_y)
        n_base, oy0, ox0 = tb * TNB, ty * TH, tx * TW
        sOff = np.full(npix, -1, dtype=np.int64)
        for p in range(npix):
            hx, r = p % HW, p // HW
            hy, b = r % HH, r // HH
            n, iy, ix = n_base + b, oy0 * stride - pad + hy, ox0 * stride - pad + hx
            if n < N and 0 <= iy < H and 0 <= ix < W:
                sOff[p] = ((n * H + iy) * W + ix) * cs_in
        for by in range(gy):
            n0 = by * TN
            acc = np.zeros((WM * WN, MT, NT, 16, 16), dtype=np.float32)
            sA = np.zeros((CKQ, npixp, 4), dtype=np.float32)
            for c in range(nchunk):
                for t0 in range(0, taps, tps):
                    if t0 == 0:
                        sA[:] = np.nan
                        for e in range(npix * CKQ):
                            q, p = e & 3, e >> 2
                            ci = c * CK + q * 4
                            v = np.zeros(4, np.float32)
                            if sOff[p] >= 0 and ci < cs_in:
                                v = xf[sOff[p] + ci: sOff[p] + ci + 4]
                            sA[q, p ^ (q << 1)] = v
                    nts = min(tps, taps - t0)
                    sB = np.zeros((nts * CKQ * TN, 4), dtype=np.float32)
                    wbase = (c * taps + t0) * CKQ * CoutP
                    for e in range(nts * CKQ * TN):
                        j, tq = e % TN, e // TN
                        if n0 + j < CoutP:
                            sB[e] = w4[wbase + tq * CoutP + n0 + j]
                    for wave in range(WM * WN):
                        wm, wn = wave // WN, wave % WN
                        for tt in range(nts):
                            t = t0 + tt
                            ky, kx = t // KW, t % KW
                            dpix = ky * HW + kx
                            af, bf = [], []
                            for mt in range(MT):
                                m = (wm * MT + mt) * 16 + li
                                b = m // tile_px
                                rem = m - b * tile_px
                                yy, xx = rem // TW, rem % TW
                                b = np.where(b >= TNB, 0, b)
                                pixbase = (b * HH + yy * stride) * HW + xx * stride
                                af.append(sA[kq, (pixbase + dpix) ^ (kq << 1)])      # [64,4]
                            for nt in range(NT):
                                bf.append(sB[(tt * CKQ + kq) * TN + (wn * NT + nt) * 16 + li])
                            for mt in range(MT):
                                for nt in range(NT):
                                    for s in range(4):
                                        mfma_16x16x4(af[mt][:, s], bf[nt][:, s], acc[wave, mt, nt])
            # epilogue
            for wave in range(WM * WN):
                wm, wn = wave // WN, wave % WN
                for mt in range(MT):
                    for lane in range(64):
                        for r in range(4):
                            m = (wm * MT + mt) * 16 + (lane >> 4) * 4 + r
                            b = m // tile_px
                            rem = m - b * tile_px
                            yy, xx = rem // TW, rem % TW
                            n, oy, ox = n_base + b, oy0 + yy, ox0 + xx
                            if b >= TNB or n >= N or oy >= Ho or ox >= Wo:
                                continue
                            for nt in range(NT):
                                co = n0 + (wn * NT + nt) * 16 + (lane & 15)
                                if co >= CoutP:
                                    continue
                                v = acc[wave, mt, nt][(lane >> 4) * 4 + r, lane & 15] * scale[co] + shift[co]
                                if out_nchw:
                                    if co < Cout:
                                        y[n, co, oy, ox] = _act(v, act_id)
                                    continue
                                if co >= cs_out:
                                    continue
                                if res is not None and not res_after:
                                    v = v + res[n, oy, ox, co]
                                v = _act(v, act_id)
                                if res is not None and res_after:
                                    v = res[n, oy, ox, co] + v
                                if co >= Cout:
                                    v = 0.0
                                y[n, oy, ox, co] = v
    return y


def _act(v, a):
    if a == 1:
        return max(v, 0.0)
    if a == 2:
        return 1.0 / (1.0 + np.exp(-v))
    if a == 3:
        return v if v > 0 else 0.01 * v
    return v
