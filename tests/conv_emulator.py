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
    dma = cfg > 10          # LDS-DMA family: pixel-major halo tile sA[p][q], no swizzle, pre-zeroed
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
        # per-lane staging offsets (element units here; bytes in the kernel), None = OOB -> zeros
        tids = np.arange(256)
        A_IT = -(-npix * CKQ // 256)
        aoff = {}
        for tid in tids:
            q, p0 = tid & 3, tid >> 2
            for it in range(A_IT):
                p = p0 + it * 64
                off = None
                if p < npix:
                    hx, r = p % HW, p // HW
                    hy, b = r % HH, r // HH
                    n, iy, ix = n_base + b, oy0 * stride - pad + hy, ox0 * stride - pad + hx
                    if n < N and 0 <= iy < H and 0 <= ix < W:
                        off = ((n * H + iy) * W + ix) * cs_in + q * 4
                aoff[(tid, it)] = off
        for by in range(gy):
            n0 = by * TN
            acc = np.zeros((WM * WN, MT, NT, 16, 16), dtype=np.float32)
            sA = np.zeros((CKQ, npixp, 4), dtype=np.float32)
            for c in range(nchunk):
                for t0 in range(0, taps, tps):
                    if t0 == 0:
                        sA[:] = 0.0 if dma else np.nan
                        for tid in tids:
                            q, p0 = tid & 3, tid >> 2
                            cpad = (c * CK + q * 4) >= cs_in
                            for it in range(A_IT):
                                p = p0 + it * 64
                                if p >= npix:
                                    continue
                                off = aoff[(tid, it)]
                                v = np.zeros(4, np.float32)
                                if off is not None and not cpad:
                                    v = xf[off + c * CK: off + c * CK + 4]
                                if dma:
                                    if off is not None and not cpad:
                                        sA[q, p] = v          # slot p*4+q; padding lanes never DMA (pre-zeroed)
                                else:
                                    sA[q, p ^ (q << 1)] = v
                    nts = min(tps, taps - t0)
                    sB = np.full((nts * CKQ * TN, 4), np.nan, dtype=np.float32)
                    wbase = (c * taps + t0) * CKQ * CoutP
                    for e in range(nts * CKQ * TN):        # e = tid + it*256
                        j, tq = e % TN, e // TN
                        sB[e] = w4[wbase + tq * CoutP + n0 + j] if n0 + j < CoutP else 0.0
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
                                af.append(sA[kq, pixbase + dpix] if dma else sA[kq, (pixbase + dpix) ^ (kq << 1)])
                            for nt in range(NT):
                                bf.append(sB[(tt * CKQ + kq) * TN + (wn * NT + nt) * 16 + li])
                            for mt in range(MT):
                                for nt in range(NT):
                                    for s in range(4):
                                        mfma_16x16x4(af[mt][:, s], bf[nt][:, s], acc[wave, mt, nt])
            # epilogue
            if not out_nchw:
                TNW = NT * 16
                SC_LD = TNW + 4
                TMb = WM * MT * 16
                sPix = np.full(TMb, -1, dtype=np.int64)
                for m in range(TMb):
                    b = m // tile_px
                    rem = m - b * tile_px
                    yy, xx = rem // TW, rem % TW
                    n, oy, ox = n_base + b, oy0 + yy, ox0 + xx
                    if b < TNB and n < N and oy < Ho and ox < Wo:
                        sPix[m] = (n * Ho + oy) * Wo + ox
                yf = y.reshape(-1)
                rf = res.reshape(-1) if res is not None else None
                for wave in range(WM * WN):
                    wm, wn = wave // WN, wave % WN
                    sC = np.full((MT * 16, SC_LD), np.nan, dtype=np.float32)
                    for lane in range(64):
                        l_i, k_q = lane & 15, lane >> 4
                        for nt in range(NT):
                            co = n0 + (wn * NT + nt) * 16 + l_i
                            sc, sh = (scale[co], shift[co]) if co < CoutP else (0.0, 0.0)
                            for mt in range(MT):
                                for r in range(4):
                                    sC[mt * 16 + k_q * 4 + r, nt * 16 + l_i] = \
                                        np.float32(acc[wave, mt, nt][k_q * 4 + r, l_i] * sc + sh)
                    C4 = TNW // 4
                    for idx in range(MT * 16 * C4):
                        row, c4 = idx // C4, idx % C4
                        pix = sPix[wm * MT * 16 + row]
                        co = n0 + wn * TNW + c4 * 4
                        if pix < 0 or co >= cs_out:
                            continue
                        v = sC[row, c4 * 4: c4 * 4 + 4].copy()
                        g0 = pix * cs_out + co
                        if rf is not None and not res_after:
                            v = v + rf[g0:g0 + 4]
                        v = np.array([_act(t, act_id) for t in v], dtype=np.float32)
                        if rf is not None and res_after:
                            v = rf[g0:g0 + 4] + v
                        for k in range(4):
                            if co + k >= Cout:
                                v[k] = 0.0
                        assert np.all(np.isnan(yf[g0:g0 + 4])), 'element stored twice'
                        yf[g0:g0 + 4] = v
                continue
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
                                if co >= Cout:
                                    continue
                                v = acc[wave, mt, nt][(lane >> 4) * 4 + r, lane & 15] * scale[co] + shift[co]
                                y[n, co, oy, ox] = _act(v, act_id)
    return y


def _act(v, a):
    if a == 1:
        return max(v, 0.0)
    if a == 2:
        return 1.0 / (1.0 + np.exp(-v))
    if a == 3:
        return v if v > 0 else 0.01 * v
    return v
