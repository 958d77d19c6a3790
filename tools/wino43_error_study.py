#!/usr/bin/env python
"""CPU study (numpy, no GPU): would Winograd F(4x4,3x3) -- 36 multiplies per 16 outputs, 1.78x fewer than the
F(2x2,3x3) the kernels use, 4x fewer than the direct sum -- hold the parity bar in fp32?

Emulates fp32 arithmetic of the three algorithms on HRNet-like layers (C channels, post-ReLU inputs, He-scaled
filters): transforms in fp32 (filter transform in float64, rounded once, as the product code does), products and
channel sums in fp32, against the float64 direct convolution.  Prints the max / rms error relative to the
output's rms for one layer and for a chain of 3x3 conv + ReLU layers (errors compound through the 100+
layers of the backbone).

    python tools/wino43_error_study.py [--channels 48] [--size 32] [--layers 8]
"""
import argparse

import numpy as np

F32 = np.float32


def mats(points):
    """Cook-Toom matrices (A^T, G, B^T) of F(m, 3) for m = len(points) + 1 - 2 finite points + infinity."""
    import numpy.polynomial.polynomial as P
    n = len(points) + 1                         # transform size alpha = m + 2
    m = n - 2
    pts = np.array(points, dtype=np.float64)
    # Vandermonde-style construction (Lavin & Gray / wincnn): A^T [m x n], G [n x 3], B^T [n x n]
    AT = np.zeros((m, n)); G = np.zeros((n, 3)); 
    for i, a in enumerate(pts):
        AT[:, i] = a ** np.arange(m)
        G[i, :] = a ** np.arange(3)
    AT[m - 1, n - 1] = 1.0
    G[n - 1, 2] = 1.0
    # scale rows of G by 1 / prod_{j != i} (a_i - a_j)
    for i, a in enumerate(pts):
        G[i, :] /= np.prod([a - b for j, b in enumerate(pts) if j != i])
    # B^T from the polynomial identities: row i = coefficients of prod_{j != i} (x - a_j), last row = prod_j (x - a_j)
    BT = np.zeros((n, n))
    for i in range(n - 1):
        c = np.array([1.0])
        for j, b in enumerate(pts):
            if j != i:
                c = P.polymul(c, np.array([-b, 1.0]))
        BT[i, :len(c)] = c
    c = np.array([1.0])
    for b in pts:
        c = P.polymul(c, np.array([-b, 1.0]))
    BT[n - 1, :len(c)] = c
    return AT, G, BT


def conv_direct(x, w, dt):
    """x [C,H,W] (zero padded by the caller to H+2, W+2), w [Co,C,3,3] -> [Co,H,W], sums in dtype dt."""
    c, hp, wp = x.shape
    h, wd = hp - 2, wp - 2
    y = np.zeros((w.shape[0], h, wd), dtype=dt)
    xs, ws = x.astype(dt), w.astype(dt)
    for ky in range(3):
        for kx in range(3):
            y += np.einsum('oc,chw->ohw', ws[:, :, ky, kx], xs[:, ky:ky + h, kx:kx + wd]).astype(dt)
    return y


def conv_wino(x, w, AT, G, BT):
    """Winograd F(m x m, 3 x 3) with fp32 data transforms / products / sums; filter transform in float64."""
    m, n = AT.shape
    c, hp, wp = x.shape
    h, wd = hp - 2, wp - 2
    assert h % m == 0 and wd % m == 0
    U = np.einsum('ia,ocab,jb->ocij', G, w.astype(np.float64), G).astype(F32)           # rounded once
    ATf, BTf = AT.astype(F32), BT.astype(F32)
    y = np.zeros((w.shape[0], h, wd), dtype=F32)
    xs = x.astype(F32)
    for ty in range(h // m):
        for tx in range(wd // m):
            d = xs[:, ty * m:ty * m + n, tx * m:tx * m + n]
            V = np.einsum('ia,cab,jb->cij', BTf, d, BTf).astype(F32)
            M = np.einsum('ocij,cij->oij', U, V).astype(F32)
            y[:, ty * m:(ty + 1) * m, tx * m:(tx + 1) * m] = np.einsum('ia,oab,jb->oij', ATf, M, ATf).astype(F32)
    return y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--channels', type=int, default=48)
    ap.add_argument('--size', type=int, default=32)
    ap.add_argument('--layers', type=int, default=8)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    C, S = a.channels, a.size
    algos = {
        'direct fp32': None,
        'F(2x2,3x3) points 0,1,-1': mats([0.0, 1.0, -1.0]),
        'F(4x4,3x3) points 0,1,-1,2,-2': mats([0.0, 1.0, -1.0, 2.0, -2.0]),
        'F(4x4,3x3) points 0,1,-1,1/2,-1/2': mats([0.0, 1.0, -1.0, 0.5, -0.5]),
    }
    # self-check of the matrices in float64
    for name, mt in algos.items():
        if mt is None:
            continue
        xx, ww = rng.standard_normal((3, S + 2, S + 2)), rng.standard_normal((2, 3, 3, 3))
        AT, G, BT = mt
        ref = conv_direct(xx, ww, np.float64)
        U = np.einsum('ia,ocab,jb->ocij', G, ww, G)
        m, n = AT.shape
        got = np.zeros_like(ref)
        for ty in range(S // m):
            for tx in range(S // m):
                V = np.einsum('ia,cab,jb->cij', BT, xx[:, ty * m:ty * m + n, tx * m:tx * m + n], BT)
                got[:, ty * m:(ty + 1) * m, tx * m:(tx + 1) * m] = np.einsum('ia,oab,jb->oij', AT, np.einsum('ocij,cij->oij', U, V), AT)
        assert np.abs(got - ref).max() < 1e-9, (name, np.abs(got - ref).max())
    ws = [rng.standard_normal((C, C, 3, 3)) * np.sqrt(2.0 / (9 * C)) for _ in range(a.layers)]
    x0 = np.maximum(rng.standard_normal((C, S, S)), 0.0)
    print('C = %d, %d x %d maps, %d layers of conv3x3 + ReLU; errors relative to the rms of the float64 output' % (C, S, S, a.layers))
    for name, mt in algos.items():
        xr, xa = x0.copy(), x0.astype(F32)
        line = []
        for li, w in enumerate(ws):
            ref = conv_direct(np.pad(xr, ((0, 0), (1, 1), (1, 1))), w, np.float64)
            xp = np.pad(xa, ((0, 0), (1, 1), (1, 1)))
            got = conv_direct(xp, w.astype(F32), F32) if mt is None else conv_wino(xp, w, *mt)
            rms = np.sqrt(np.mean(ref ** 2))
            if li in (0, a.layers - 1):
                line.append('layer %d: max %.2e rms %.2e' % (li + 1, np.abs(got - ref).max() / rms,
                                                             np.sqrt(np.mean((got - ref) ** 2)) / rms))
            xr, xa = np.maximum(ref, 0.0), np.maximum(got, F32(0))
        print('%-36s %s' % (name, '   '.join(line)))


if __name__ == '__main__':
    main()
