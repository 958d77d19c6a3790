"""conv_wino4_kernel (csrc/conv_wino4.hip) without a GPU: the algebra the kernel implements restated in numpy
(B^T d B, A^T M A with the kernel's own instruction forms), the filter layout its waves load, the halo slot order
that keeps the transform's LDS reads off each other's banks, and -- on the ISA hipcc produces here -- that no
register of an asynchronous filter load is touched before its s_waitcnt (tools/check_wino4_isa.py)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
               [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
              [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)


def bt6(t):          # w4_bt: the 12-instruction form
    o = np.empty(6)
    o[0] = 4 * t[0] + (-5 * t[2] + t[4])
    u, v = -4 * t[2] + t[4], -4 * t[1] + t[3]
    o[1], o[2] = u + v, u - v
    p, q = t[4] - t[2], t[3] - t[1]
    o[3], o[4] = 2 * q + p, -2 * q + p
    o[5] = 4 * t[1] + (-5 * t[3] + t[5])
    return o


def at6(m):          # w4_at: the 10-instruction form
    p, q, r, s = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
    return np.array([m[0] + p + r, 2 * s + q, 4 * r + p, 8 * s + q + m[5]])


def test_instruction_forms_equal_the_matrices():
    rng = np.random.default_rng(0)
    for _ in range(20):
        t = rng.standard_normal(6)
        np.testing.assert_allclose(bt6(t), BT @ t, atol=1e-12)
        np.testing.assert_allclose(at6(t), AT @ t, atol=1e-12)


def test_thirds_of_the_input_transform_cover_b_t_d_b():
    """PART 0 rows {0, 5}, PART 1 rows {1, 2}, PART 2 rows {3, 4} of the row pass, each followed by the column pass."""
    rng = np.random.default_rng(1)
    d = rng.standard_normal((6, 6))
    V = np.empty((6, 6))
    V[0] = bt6(4 * d[0] - 5 * d[2] + d[4])
    V[5] = bt6(4 * d[1] - 5 * d[3] + d[5])
    u, v = d[4] - 4 * d[2], d[3] - 4 * d[1]
    V[1], V[2] = bt6(u + v), bt6(u - v)
    p, q = d[4] - d[2], d[3] - d[1]
    V[3], V[4] = bt6(p + 2 * q), bt6(p - 2 * q)
    np.testing.assert_allclose(V, BT @ d @ BT.T, atol=1e-12)


def test_f43_convolution_identity_and_packed_filter_layout():
    """Y = A^T [ sum_ci (G g G^T) * (B^T d B) ] A equals the direct 3x3 correlation on a 4x4 output tile, with U read
    back out of engine.pack_wino4_weight's layout exactly as a wave's lane does."""
    from egonet_amd import engine
    rng = np.random.default_rng(2)
    cout, cin = 48, 16
    w = rng.standard_normal((cout, cin, 3, 3))
    x = rng.standard_normal((cin, 6, 6))
    U = engine.pack_wino4_weight(torch.from_numpy(w)).numpy().reshape(cout // 48, cin // 8, 2, 12, 3, 64, 4)
    M = np.zeros((cout, 6, 6))
    for ci in range(cin):
        Vt = BT @ x[ci] @ BT.T
        stage, g, kq = ci // 8, (ci % 8) // 4, ci % 4
        for co in range(cout):
            nt, li = co // 16, co % 16
            for pt in range(36):
                wave, pl = pt // 3, pt % 3
                p_ = 3 * pl + nt
                u = U[0, stage, g, wave, p_ // 4, 16 * kq + li, p_ % 4]
                assert abs(u - (G @ w[co, ci] @ G.T)[pt // 6, pt % 6]) < 1e-6
                M[co, pt // 6, pt % 6] += u * Vt[pt // 6, pt % 6]
    Y = np.einsum('ai,cij,bj->cab', AT, M, AT)
    ref = np.zeros((cout, 4, 4))
    for a in range(4):
        for b in range(4):
            ref[:, a, b] = np.einsum('ocij,cij->o', w, x[:, a:a + 3, b:b + 3])
    np.testing.assert_allclose(Y, ref, rtol=1e-5, atol=2e-4)      # U is rounded to fp32 once
    # padding values of the dwordx4 layout are zero
    assert float(np.abs(U[..., 2, :, 1:]).max()) == 0.0


def test_halo_slot_order_is_bank_conflict_free_for_the_transform_reads():
    """8-byte slot of pixel (y, x), channel pair p: p * 721 + (x % 4) * 180 + y * 10 + x // 4 (the odd pair pitch is the
    bank skew of the halo STORES, see test_halo_stores_are_bank_conflict_free).  Lane (tile (ty, tx) of
    an m-tile, channel kq of k-group g) reads dword kq & 1 of pair 2 g + kq // 2 at pixel (4 ty + i, 4 tx + j).  The
    hardware banks ds_read_b32 per 32-lane group on dword mod 32 (MI355X_MICROARCH.md): for every (i, j) each group
    must hit 32 different banks; every (pixel, pair) has exactly one slot below 2884."""
    def slot(p, y, x):
        return p * 721 + (x % 4) * 180 + y * 10 + x // 4
    seen = set()
    for p in range(4):
        for y in range(18):
            for x in range(34):
                s = slot(p, y, x)
                assert 0 <= s < 2884 and s not in seen
                seen.add(s)
    for mt in range(2):
        for g in range(2):
            for i in range(6):
                for j in range(6):
                    for half in (range(0, 32), range(32, 64)):
                        banks = set()
                        for lane in half:
                            li, kq = lane & 15, lane >> 4
                            ty, tx = 2 * mt + (li >> 3), li & 7
                            dword = slot(2 * g + kq // 2, 4 * ty + i, 4 * tx + j) * 2 + (kq & 1)
                            banks.add(dword % 32)
                        assert len(banks) == 32, (mt, g, i, j, len(banks))


def test_halo_slot_order_of_the_16x16_region_geometry():
    """conv_wino4b_kernel (GEO 1): 18 x 18 halo pixels x 16 channels = 8 channel-pair planes,
    slot = p * 394 + (x % 4) * 97 + y * 5 + x // 4 (plane / pair pitches padded for the stores' banks).  Lane (tile
    (ty, tx) of the 4 x 4 tiles, channel kq of k-group g) reads
    dword kq & 1 of pair 2 g + kq // 2 at pixel (4 ty + i, 4 tx + j): dword = 40 ty + 2 tx + (kq & 1) + const -- ty adds
    8 banks, so every 32-lane group hits 32 different banks for every (i, j); the load lane of element
    e = (wave + 12 k) * 64 + lane fetches element e // 4 of the 18 x 20 (row-padded) pixel list, channel quad e % 4, and
    stores pairs 2 q, 2 q + 1."""
    def slot(p, y, x):
        return p * 394 + (x % 4) * 97 + y * 5 + x // 4
    seen = set()
    for p in range(8):
        for y in range(18):
            for x in range(18):
                s = slot(p, y, x)
                assert 0 <= s < 3152 and s not in seen
                seen.add(s)
    assert len(seen) == 8 * 18 * 18
    for g in range(4):
        for i in range(6):
            for j in range(6):
                for half in (range(0, 32), range(32, 64)):
                    banks = set()
                    for lane in half:
                        li, kq = lane & 15, lane >> 4
                        ty, tx = li >> 2, li & 3
                        banks.add((slot(2 * g + kq // 2, 4 * ty + i, 4 * tx + j) * 2 + (kq & 1)) % 32)
                    assert len(banks) == 32, (g, i, j, len(banks))
    # the two load pieces of the 12 waves cover every (pixel, quad) exactly once
    got = set()
    for k in range(2):
        for wave in range(12):
            for lane in range(64):
                e = (wave + 12 * k) * 64 + lane
                px, q = e // 4, e % 4
                hy, hx = divmod(px, 20)
                if hy < 18 and hx < 18:
                    assert (hy, hx, q) not in got
                    got.add((hy, hx, q))
    assert len(got) == 18 * 18 * 4


def test_halo_slot_order_of_the_four_image_geometry():
    """conv_wino4c_kernel (GEO 2): four 8 x 8 images per region, 10 x 10 halo pixels each x 16 channels = 8 channel-pair
    planes, slot = p * 586 + (x % 4) * 145 + imgbase(img) + y * 3 + x // 4 with imgbase = 34 img + 4 (img // 2).  Lane
    (tile li: image li // 4, tile (ty, tx) = ((li // 2) % 2, li % 2) of its 2 x 2 tiles; channel kq of k-group g) reads
    dword kq & 1 of pair 2 g + kq // 2 at pixel (4 ty + i, 4 tx + j): the 16 tiles fall on 16 different even dwords mod
    32, so every 32-lane group hits 32 banks.  Three load pieces per wave cover the 4 x 10 x 12 (row-padded) pixel list x
    4 channel quads; their ds_write_b64 (groups of 16 lanes = 4 aligned pixels x 4 quads) hit 32 banks too.  The K split
    (KS = 2) hands stages [ks S, (ks + 1) S) of the pixel's channels and k-groups [48 ks, 48 ks + 48) of the filter to
    item half ks."""
    XD, PLANE, PAIR, QPP, RWP = 3, 145, 586, 4, 12
    def imgbase(img):
        return 34 * img + 4 * (img >> 1)
    def slot(p, img, y, x):
        return p * PAIR + (x & 3) * PLANE + imgbase(img) + y * XD + (x >> 2)
    seen = set()
    for p in range(8):
        for img in range(4):
            for y in range(10):
                for x in range(10):
                    s = slot(p, img, y, x)
                    assert 0 <= s < 8 * PAIR and s not in seen
                    seen.add(s)
    assert 8 * PAIR * 8 + 1024 <= 38 * 1024 and 2 * 36 * 1024 + 2 * 38 * 1024 + 12 * 96 * 8 <= 160 * 1024 - 512
    for g in range(4):
        for i in range(6):
            for j in range(6):
                for half in (range(0, 32), range(32, 64)):
                    banks = set()
                    for lane in half:
                        li, kq = lane & 15, lane >> 4
                        img, ty, tx = li >> 2, (li >> 1) & 1, li & 1
                        banks.add((slot(2 * g + kq // 2, img, 4 * ty + i, 4 * tx + j) * 2 + (kq & 1)) % 32)
                    assert len(banks) == 32, (g, i, j, len(banks))
    got, full = set(), 0
    for k in range(3):
        for wave in range(12):
            for grp in range(4):
                for second in (0, 1):
                    banks, valid = [], 0
                    for lane in range(grp * 16, grp * 16 + 16):
                        e = (wave + 12 * k) * 64 + lane
                        px, hq = divmod(e, QPP)
                        hya, hx = divmod(px, RWP)
                        img, hy = divmod(hya, 10)
                        if img < 4 and hx < 10:
                            valid += 1
                            if second == 0:
                                assert (img, hy, hx, hq) not in got
                                got.add((img, hy, hx, hq))
                            sl = slot(2 * hq + second, img, hy, hx)
                            banks += [(2 * sl) % 32, (2 * sl + 1) % 32]
                    if valid == 16:
                        full += 1
                        assert len(set(banks)) == 32, (k, wave, grp)
                    else:
                        assert len(set(banks)) == len(banks), (k, wave, grp)
    assert len(got) == 4 * 10 * 10 * 4 and full >= 160
    # output tile of lane (wave, lane) in the exchange round: tile = 4 (wave & 3) + (lane >> 4) -> (image, ty, tx)
    tiles = {(t >> 2, (t >> 1) & 1, t & 1) for t in range(16)}
    assert len(tiles) == 16
    # K split of 384 input channels: 24 stages of 16 channels -> 12 per half; k-groups of 4 channels -> 48 per half
    cin, ks_n = 384, 2
    s_half = cin // (16 * ks_n)
    for ks in range(ks_n):
        chans = [ks * s_half * 16 + s * 16 + c for s in range(s_half) for c in range(16)]
        groups = [ks * (cin // 4) // ks_n + 4 * s + h for s in range(s_half) for h in range(4)]
        assert chans == list(range(ks * cin // 2, (ks + 1) * cin // 2))
        assert [4 * h_ + q for h_ in groups for q in range(4)] == chans


def test_item_orders_visit_every_region_cotile_and_half_once():
    """w4_item_mode / w4_item_count and the item decode of w4_body, restated: block w -> (xcd = w & 7, q = w >> 3) ->
    (region, co-tile, K half).  Mode 0 (1 - 3, 6 co-tiles) puts 8 consecutive regions on the 8 XCDs; mode 1 (4
    co-tiles: the 192-channel branch) co-tile xcd % nct with every (8 / nct)-th (region, half) per XCD; mode 2 (multiples
    of 8: the 384-channel branch) co-tile 8 j + xcd.  Every (region, co-tile, half) exactly once, padding items are
    skipped; in modes 0 and 2 both halves of a pair land on the same XCD (a matter of speed, never of correctness)."""
    def mode(nct):
        if nct % 8 == 0:
            return 2
        return 1 if nct == 4 else 0          # (W4_COX_MIN_NCT = 4)
    def count(m, nreg, nct, ks):
        if m == 2:
            return nreg * nct * ks
        if m == 1:
            return 8 * ((nreg * ks + 8 // nct - 1) // (8 // nct))
        return ((nreg + 7) >> 3) * nct * ks * 8
    for nct in (1, 2, 3, 4, 6, 8, 16):
        for KS in (1, 2):
            for nreg in (1, 3, 4, 16, 17, 64):
                m = mode(nct)
                seen = {}
                for w in range(count(m, nreg, nct, KS)):
                    xq, q = w & 7, w >> 3
                    if m == 0:
                        qq = q // (nct * KS)
                        cs = q - qq * nct * KS
                        reg, ct, ks = qq * 8 + xq, cs // KS, cs % KS
                    elif m == 1:
                        lg = nct >> 1
                        cs = q * (8 >> lg) + (xq >> lg)
                        reg, ct, ks = cs // KS, xq & (nct - 1), cs % KS
                    else:
                        c8 = nct >> 3
                        qq = q // c8
                        reg, ct, ks = qq // KS, (q - qq * c8) * 8 + xq, qq % KS
                    if reg >= nreg:
                        continue
                    assert 0 <= ct < nct and (reg, ct, ks) not in seen
                    seen[(reg, ct, ks)] = xq
                assert len(seen) == nreg * nct * KS, (nct, KS, nreg, m)
                if KS == 2 and m != 1:      # (mode 1 spreads a pair over two XCDs: the hand-off is placement-independent)
                    assert all(seen[(r, c, 0)] == seen[(r, c, 1)] for r in range(nreg) for c in range(nct))


def test_every_item_of_a_block_has_the_same_cotile():
    """conv_wino4s_kernel (the training tape's build, round 5) writes ONE BatchNorm partial row per block that covers one
    co-tile: with the launcher's grid (w4_grid: whole XCD rounds of (co-tile, K half) up to one block per CU) the items
    w, w + grid, ... of a block decode to the same co-tile in all three item orders."""
    def mode(nct):
        if nct % 8 == 0:
            return 2
        return 1 if nct == 4 else 0
    def count(m, nreg, nct, ks):
        if m == 2:
            return nreg * nct * ks
        if m == 1:
            return 8 * ((nreg * ks + 8 // nct - 1) // (8 // nct))
        return ((nreg + 7) >> 3) * nct * ks * 8
    def cotile(m, w, nct, KS):
        xq, q = w & 7, w >> 3
        if m == 0:
            return (q % (nct * KS)) // KS
        if m == 1:
            return xq & (nct - 1)
        c8 = nct >> 3
        return (q % c8) * 8 + xq
    for cus in (256, 304, 64, 8):
        for nct in (1, 2, 3, 4, 6, 8, 16):
            for KS in (1, 2):
                nck = nct * KS
                for nreg in (1, 5, 16, 64, 512, 2048):
                    m = mode(nct)
                    nwork = count(m, nreg, nct, KS)
                    cap = cus // (8 * nck) * (8 * nck) or 8 * nck
                    grid = min(nwork, cap)
                    for b in range(grid):
                        cts = {cotile(m, w, nct, KS) for w in range(b, nwork, grid)}
                        assert len(cts) == 1, (cus, nct, KS, nreg, b, cts)


def test_ticket_hand_off_of_the_k_split_in_every_interleaving():
    """The protocol of w4_body's K-split item end as a state machine over its atomic steps, run under every schedule of
    the two blocks of a pair: block = draw a ticket (fetch_add on the pair's word); ticket 0 -> store the share, bump the
    word; any other ticket (1 if the partner has not bumped yet, 2 if it has) -> poll until the word is 3, read the share,
    write the result, reset the word to 0.  Claims: the waiting block only ever waits for a partner that HAS drawn its
    ticket (no dependence on dispatch order), the share is read only after it was stored, exactly one block writes the
    result, and the word is 0 again at the end (the next launch starts from zero)."""
    import itertools

    def block(name, m):
        t = m['word']
        m['word'] += 1                                   # draw
        yield
        if t == 0:
            m['share'] = name                            # sc1 stores, drained
            yield
            m['word'] += 1                               # bump: -> 2 (partner still to draw) or 3
            yield
        else:
            assert t in (1, 2), t
            polls = 0
            while m['word'] != 3:                        # t != 0: the partner drew first, so it is in its item end
                polls += 1
                assert polls < 64, 'waits for a block that makes no progress'
                yield
            assert m['share'] not in (None, name)        # reads what the OTHER block stored
            m['result'].append(name)
            yield
            m['word'] = 0
            yield

    tickets_seen = set()
    for launches in (1, 2):                              # (two launches in a row on the same word)
        for sched in itertools.product((0, 1), repeat=9):
            m = dict(word=0, share=None, result=[])
            for _ in range(launches):
                m['share'], m['result'] = None, []
                gens = [block('a', m), block('b', m)]
                done = [False, False]
                order = list(sched) + [0, 1] * 40               # the schedule, then a fair tail: both blocks are RUNNING
                for k in order:
                    if done[k]:
                        continue
                    w0 = m['word']
                    try:
                        next(gens[k])
                    except StopIteration:
                        done[k] = True
                    if m['word'] == w0 + 1 and w0 in (1, 2) and m['share'] is None:
                        tickets_seen.add(w0)
                assert all(done) and m['word'] == 0 and len(m['result']) == 1, (sched, m)
    assert tickets_seen <= {1, 2}


@pytest.mark.parametrize('geo', [0, 1])
def test_halo_stores_are_bank_conflict_free(geo):
    """The halo goes global -> registers -> LDS: lane e of a load piece holds 16 bytes (a channel quad) of one pixel and
    stores its two channel pairs with two ds_write_b64 (pair planes 2 q and 2 q + 1).  ds_write_b64 is served in groups of
    16 contiguous lanes, bank = dword mod 32 (MI355X_MICROARCH.md): a group = 8 aligned pixels x 2 quads (GEO 0) or 4
    aligned pixels x 4 quads (GEO 1) -- the row pitch of the element list is padded to 40 / 20 for that -- must hit 32
    different banks.  Round 3's pitches (pair 720 / 360, plane 180 / 90, unpadded rows) put the quads of a pixel on ONE
    bank: 2- / 4-way conflicts on every halo store (SQ_LDS_BANK_CONFLICT 1.26 M cycles per launch)."""
    RH, RW, RWP, XD = 18, (18 if geo else 34), (20 if geo else 40), (5 if geo else 10)
    PLANE, PAIR, QPP = (97 if geo else 180), (394 if geo else 721), (4 if geo else 2)
    full = 0
    for k in range(2):
        for wave in range(12):
            for grp in range(4):
                for second in (0, 1):
                    banks, valid = [], 0
                    for lane in range(grp * 16, grp * 16 + 16):
                        e = (wave + 12 * k) * 64 + lane
                        px, hq = divmod(e, QPP)
                        hy, hx = divmod(px, RWP)
                        if hy < RH and hx < RW:
                            valid += 1
                            sl = (2 * hq + second) * PAIR + (hx & 3) * PLANE + hy * XD + (hx >> 2)
                            banks += [(2 * sl) % 32, (2 * sl + 1) % 32]
                    if valid == 16:
                        full += 1
                        assert len(set(banks)) == 32, (geo, k, wave, grp)
                    else:                       # row tails / lanes past the halo (parked): the valid lanes still never collide
                        assert len(set(banks)) == len(banks), (geo, k, wave, grp)
    assert full >= (136 if geo else 140)


def test_exchange_rotation_makes_the_output_reads_bank_conflict_free():
    """Item end: writer lane (li_w, kq_w) parks its C fragment (tiles 4 kq_w .. + 3 of channel li_w) as a float4 in slot
    `lane` of [point][co sub-tile][64 slots]; reader lane (li, kq) of wave (ont, okq) takes tile 4 okq + kq = element kq
    of slot 16 okq + li.  Plain float4 slots: dword 4 li + kq -> li and li + 8 share a bank (2-way on all 36 reads).
    Writers with li_w >= 8 rotate their float4 by two dwords, the reader looks at dword (kq + 2 (li >> 3)) & 3: the 32
    lanes of a ds_read_b32 group hit 32 banks, and every reader still gets tile 4 okq + kq."""
    store = {}
    for lane_w in range(64):
        li_w, kq_w = lane_w & 15, lane_w >> 4
        rot = 2 * (li_w >> 3)
        for r in range(4):
            store[(lane_w, (r + rot) & 3)] = (4 * kq_w + r, li_w)        # dword position -> (tile, channel)
    for okq in range(4):
        for half in (range(0, 32), range(32, 64)):
            banks = set()
            for lane in half:
                li, kq = lane & 15, lane >> 4
                pos = (kq + 2 * (li >> 3)) & 3
                slot_ = 16 * okq + li
                assert store[(slot_, pos)] == (4 * okq + kq, li)
                banks.add((4 * slot_ + pos) % 32)
            assert len(banks) == 32
    # the writers' two ds_write_b64 (16-lane groups): elements (0, 1) at +8 * hi, (2, 3) at +8 - 8 * hi
    for grp in range(4):
        for first in (True, False):
            banks = []
            for lane in range(grp * 16, grp * 16 + 16):
                hi = (lane & 15) >> 3
                off = (2 * hi if first else 2 - 2 * hi)
                banks += [(4 * lane + off) % 32, (4 * lane + off + 1) % 32]
            assert len(set(banks)) == 32


def test_wino4b_reads_the_same_filter_pack_in_k_group_pairs():
    """GEO 1 walks engine.pack_wino4_weight's layout as [co-tile][k-group h = Cin / 4][wave][3 x 64 x 4]: stage s (16
    channels) = k-groups 4 s .. 4 s + 3, wait group G = the pair (4 s + 2 G, 4 s + 2 G + 1); the byte offset of k-group
    h is h * 36 864 -- the same bytes GEO 0 addresses as [stage = Cin / 8][g]."""
    from egonet_amd import engine
    rng = np.random.default_rng(5)
    cout, cin = 48, 32
    w = rng.standard_normal((cout, cin, 3, 3))
    flat = engine.pack_wino4_weight(torch.from_numpy(w)).numpy()
    a = flat.reshape(cout // 48, cin // 8, 2, 12, 3, 64, 4)                 # GEO 0 view
    b = flat.reshape(cout // 48, cin // 4, 12, 3, 64, 4)                    # GEO 1 view: linear k-group index
    for h in range(cin // 4):
        assert np.array_equal(b[0, h], a[0, h // 2, h % 2])
    assert flat.size == (cout // 48) * (cin // 4) * 9216
    # channel of lane (kq) in k-group h: ci = 4 h + kq
    U = np.einsum('ia,ocab,jb->ocij', G, w, G)
    for h in (0, 3, 7):
        for wave in (0, 5, 11):
            for p_ in range(9):
                pl, nt = p_ // 3, p_ % 3
                for kq in range(4):
                    for li in (0, 7, 15):
                        got = b[0, h, wave, p_ // 4, 16 * kq + li, p_ % 4]
                        pt = 3 * wave + pl
                        assert abs(got - U[16 * nt + li, 4 * h + kq, pt // 6, pt % 6]) < 1e-6


def test_device_filter_transform_addresses_the_register_feed_layout():
    """wino4_pack_weight_kernel (egn_wino4_pack_weight_f32) restated: thread (o, i) writes U[pt] = (G g G^T)[pt // 6][pt % 6]
    to ((ct * (n_in / 4) + i / 4) * 9216 + wave * 768 + (p / 4) * 256 + (16 (i % 4) + o % 16) * 4 + p % 4 with wave = pt / 3,
    p = 3 (pt % 3) + nt, nt = (o % 48) / 16; the thread with nt = 0 zeroes values 9..11.  dgrad swaps the channels and
    rotates the taps.  Must equal engine.pack_wino4_weight (the layout the kernels are tested with) element for element."""
    from egonet_amd import engine
    rng = np.random.default_rng(0)
    for dgrad in (0, 1):
        cout, cin = (96, 48) if not dgrad else (48, 96)
        w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
        n_out, n_in = (cin, cout) if dgrad else (cout, cin)
        dst = np.full((n_out // 48) * (n_in // 4) * 9216, np.nan, np.float32)
        for o in range(n_out):
            for i in range(n_in):
                g = (w[i, o, ::-1, ::-1] if dgrad else w[o, i]).astype(np.float64)
                U = G @ g @ G.T
                ct, nt, li = o // 48, (o % 48) >> 4, o & 15
                base = (ct * (n_in >> 2) + (i >> 2)) * 9216 + (16 * (i & 3) + li) * 4
                for pt in range(36):
                    p = 3 * (pt % 3) + nt
                    dst[base + (pt // 3) * 768 + (p >> 2) * 256 + (p & 3)] = U[pt // 6, pt % 6]
                if nt == 0:
                    for wave in range(12):
                        dst[base + wave * 768 + 513: base + wave * 768 + 516] = 0
        wt = torch.from_numpy(w)
        if dgrad:
            wt = wt.permute(1, 0, 2, 3).flip(2, 3).contiguous()
        ref = engine.pack_wino4_weight(wt).numpy()
        assert not np.isnan(dst).any() and np.abs(dst - ref).max() < 1e-6


@pytest.mark.skipif(shutil.which(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')) is None, reason='hipcc not installed')
def test_filter_load_registers_reach_their_waitcnt_untouched():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_wino4_isa.py')], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ' 0 problems' in r.stdout
