"""The algebra of csrc/conv_wgrad_wino.hip restated in numpy (no GPU): the weight gradient of a 3x3 / stride 1 /
pad 1 convolution as Winograd F(2x2,3x3),

    dU_f[co][ci] = sum_tiles dM_f[t][co] V_f[t][ci],   V = B^T d B,   dM = A dY A^T,   dg = G^T dU G,

organised the way the kernel organises it: eight "waves" (frequency row i, column pair jb), each reading only the
patch rows / columns its frequencies touch with ONE sign per direction (both transforms negate frequency 3),
folding its two columns into the two values (q0, q1) the three tap columns need, and the fixed-order sum over
the waves with the coefficients G[i][tap row] and the (q, sign) table of the source wave's jb.  Pinned against
torch's conv2d backward in float64."""
import numpy as np
import torch
import torch.nn.functional as F

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)


def wgrad_wino(x, dy):
    """x [N,Ci,H,W], dy [N,Co,H,W] (H, W even) -> dg [Co,Ci,3,3], the kernel's organisation."""
    n, ci, h, w = x.shape
    co = dy.shape[1]
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    parked = {}                                            # (i, jb) -> (q0, q1), each [Co, Ci]
    for i in range(4):
        ra, rb = [(0, 2), (1, 2), (2, 1), (3, 1)][i]       # patch rows (a, b); T = a + sr b
        sr = 1.0 if i == 1 else -1.0
        ea, eb = (1, 0) if i == 3 else (0, 1)              # dy rows (a, b); R = a + tr b
        tr = {0: 0.0, 1: 1.0, 2: -1.0, 3: 0.0}[i]
        for jb in range(2):
            ca, cb, cc = (0, 2, 1) if jb == 0 else (3, 1, 2)
            sc = 1.0 if jb == 0 else -1.0
            da, db = (0, 1) if jb == 0 else (1, 0)
            acc = np.zeros((2, co, ci))
            for ty in range(h // 2):
                for tx in range(w // 2):
                    d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]          # [N,Ci,4,4]
                    e = dy[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2]          # [N,Co,2,2]
                    T = d[:, :, ra, :] + sr * d[:, :, rb, :]                    # [N,Ci,4]
                    V = np.stack([T[..., ca] - T[..., cb], T[..., cc] + sc * T[..., cb]])      # [2,N,Ci]
                    R = e[:, :, ea, :] + tr * e[:, :, eb, :]                    # [N,Co,2]
                    M = np.stack([R[..., da], R[..., db] + sc * R[..., da]])    # [2,N,Co]
                    acc += np.einsum('fnc,fnd->fcd', M, V)
            h_ = 0.5 * acc[1]
            parked[(i, jb)] = (acc[0] + h_, h_)
    dg = np.zeros((co, ci, 3, 3))
    for ta in range(3):
        for tb in range(3):
            s = np.zeros((co, ci))
            for i in range(4):
                g = G[i, ta]
                if g == 0:
                    continue
                v0 = parked[(i, 0)][0 if tb == 0 else 1]                        # jb = 0: (q0, q1, q1)
                v1 = parked[(i, 1)][0 if tb == 2 else 1]                        # jb = 1: (q1, -q1, q0)
                s += g * (v0 - v1 if tb == 1 else v0 + v1)
            dg[:, :, ta, tb] = s
    return dg


def test_winograd_weight_gradient_organisation_equals_conv2d_backward():
    rng = np.random.default_rng(3)
    for n, ci, co, h, w in ((2, 3, 4, 4, 6), (1, 5, 2, 8, 2)):
        x, dy = rng.standard_normal((n, ci, h, w)), rng.standard_normal((n, co, h, w))
        wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(torch.from_numpy(x), wt, None, 1, 1).backward(torch.from_numpy(dy))
        np.testing.assert_allclose(wgrad_wino(x, dy), wt.grad.numpy(), rtol=0, atol=1e-11)


def test_lds_pixel_stride_is_conflict_free():
    """56 dwords per pixel: the four tiles of a K step (two pixels apart) and the 16 lanes of a group
    (consecutive channels) hit 64 distinct banks of a ds_read_b32 for every patch position."""
    for pos in range(0, 56 * 18 * 10, 56):                # any patch-position offset (a pixel multiple)
        for j in range(3):
            banks = {(pos + kq * 2 * 56 + li + 16 * j) % 64 for kq in range(4) for li in range(16)}
            assert len(banks) == 64
    # the unpadded 48-dword stride would put tiles kq and kq + 2 on the same banks
    assert len({(kq * 2 * 48 + li) % 64 for kq in range(4) for li in range(16)}) == 32
