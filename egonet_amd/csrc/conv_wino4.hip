// conv_wino4.hip -- fused Winograd F(4x4,3x3), second design: "transform once, multiply from LDS" [round 3].
//
// Why a second F(4x4,3x3) kernel.  F(4x4,3x3) executes 36 multiplies per 16 outputs and (ci, co) pair -- 1.78x
// fewer MFMAs than the F(2x2,3x3) kernels of conv_wino.hip, which sit at 61-64 % of MFMA issue in their own
// cycles (profiles/r3_wino9_kq2_timeline.txt) with nothing left to remove.  conv_wino43_kernel (the first
// attempt) had every wave transform its own frequency row in front of 18 MFMAs: 2.9 VALU + 13 SALU per MFMA, a
// barrier per 4 channels, filter stages filling the LDS -- as fast as F(2x2,3x3), no faster.  What round 3
// measured (profiles/r3_mfma_tax.txt): beside fp32 MFMAs an LDS read, an LDS-DMA piece or a global load costs
// nothing, every VALU instruction costs 3-6 cycles of matrix-pipe time.  So:
//   * the input transform V = B^T d B runs ONCE per (tile, channel) -- 144 VALU per 64 (tile, channel) pairs,
//     written to LDS as MFMA A operands (1.33 VALU per MFMA at a 48-channel co-tile) -- in thirds (frequency rows
//     {0,5} / {1,2} / {3,4}: 48 VALU each), every wave one third per stage, each third at a different point of the
//     stage so that two of a SIMD's three waves always have MFMAs to issue;
//   * the multiplying waves read A from LDS (free) and B -- the transformed filter -- straight from global
//     memory into registers, each element by exactly one wave of the block (free, and the LDS holds no
//     filter: 8-channel stages, one barrier per 36 MFMAs of a wave);
//   * 12 waves = 3 per SIMD; wave w owns frequency points 3w .. 3w+2 for all 3 co sub-tiles and both m-tiles:
//     18 MFMAs per 4-channel k-group, 72 accumulator registers.
//
// Block = 16 x 32 output pixels of one image = 4 x 8 tiles of 4 x 4 pixels = 2 m-tiles (tile rows {0,1} / {2,3})
// x 48 output channels; K stages of 8 channels (two k-groups of 4).  LDS (124 KB): two halo buffers
// [18 x 34 pixels][8 ch] (through registers: buffer_load_dwordx4 -> ds_write_b128, two stages ahead), two V buffers
// [36 points][m-tile][k-group][64] floats.  Stage s: every wave loads its 2 halo pieces of stage s + 2 and its
// 2 x 3 filter dwordx4 of the next k-groups, turns its third of halo s + 1 into V s + 1 and multiplies V s; one
// barrier.  Every vector-memory wait is vmcnt(0) -- see the note in front of W4_STAGE.
// Item end: the accumulators go through LDS once per m-tile ([point][co sub-tile][lane] float4, 108 KB) so
// that every lane gets all 36 points of ONE (tile, co): Y = A^T M A (100 VALU), scale / shift / residual / ReLU,
// 16 stores of 64 B segments.
//
// Matrices (Lavin & Gray, points 0, +-1, +-2, inf), filter transform on the host (engine.pack_wino4_weight):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Two geometries of the same body (template parameter GEO) [round 4]:
//   GEO 0  conv_wino4_kernel   16 x 32 pixel regions (2 m-tiles of 16 tiles), 8-channel stages  -- as described above;
//   GEO 1  conv_wino4b_kernel  16 x 16 pixel regions (ONE m-tile), 16-channel stages: the same 36 MFMAs, 48 transform
//          instructions and 2 halo pieces per wave and stage, the same filter layout (a stage reads four consecutive
//          k-groups instead of two) and LDS budget, but twice as many items per layer -- the 16 x 16 maps of the
//          192-channel branch are 256 items at 64 crops (two-image regions would be 128: half the chip), and small
//          batches (BASELINE configs[4]'s 16-crop shard) fill the CUs.  A "k-group pair" (8 channels x 1 m-tile) takes
//          the place of GEO 0's k-group (4 channels x 2 m-tiles) in the stage schedule.
//   GEO 2  conv_wino4c_kernel  FOUR IMAGES of an 8 x 8 map per region (the 384-channel branch: 2 x 2 tiles per image,
//          one m-tile = 4 images), 16-channel stages as GEO 1; 3 halo pieces per wave and stage (4 x 10 x 10 pixels).
//          (conv_wino4bk_kernel: GEO 1 with the same K split -- small batches: 16 crops x 192 channels are 64 items.)
//          Template parameter KS = 2 splits the input channels of an item over two blocks (64 crops x 384 -> 384
//          channels are 16 regions x 8 co-tiles = 128 items: half the chip; 256 with the split): both add their
//          raw Y = A^T M A into a zeroed y (buffer_atomic_add_f32 -- two addends onto 0: the sum does not depend on
//          the order), and conv_wino4_finish_kernel applies scale / shift / residual / ReLU in place.
// Reference: the 3x3 stride-1 convolutions of libs/model/heatmapModel/hrnet.py (BasicBlock :49-76 and the
// branches built from it); tolerance as for the F(2x2,3x3) kernels, see tools/wino43_network_study.py.
#include "conv_wino4.h"

// ABL != 0: timing ablations (WRONG RESULTS; probe builds only -- -DEGN_PROBES, tools/wino_probe.py): bit 0 no input
// transform, bit 1 no MFMAs, bit 2 no exchange / output transform / stores, bit 3 no filter loads, bit 4 no halo DMA,
// bit 5 bank-conflict-free halo reads, bit 6 s_memtime stamps (tools/wino4_clk.py).
// (Round 4 also measured two other stage schedules -- the last transform third moved to mid-stage: no change; every
// wave weaving its third between its own MFMA groups: 20-35 % slower -- profiles/r4_wino4_experiments.txt.)
// ST [round 5]: the training tape's build -- BatchNorm batch statistics in the item end (the sums / sums of squares of
// what the lane stores, per block and output channel, as conv_wino9_kernel writes them: egn_bn_stats_finalize_f32 adds
// the rows of all blocks in a fixed order).  A block's items all have the same co-tile (the grid is a multiple of
// 8 x co-tiles x KS, tests/test_wino4_design_cpu.py), so its row covers 48 channels and is zero elsewhere.
template <int ABL, int GEO, int KS = 1, bool ST = false>
__device__ __forceinline__ void w4_body(const ConvArgs& a) {
  typedef W4G<GEO> Q;
  extern __shared__ float4 w4_smem[];
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_w4_t)w4_smem;
  const float* smf = reinterpret_cast<const float*>(w4_smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int tpart = wave >> 2;             // its third of the frequency rows of the input transform
  const int tw = wave & 3;                 // its share: GEO 0 m-tile tw >> 1, k-group tw & 1; GEO 1 k-group tw

  const int C = a.Cin, Co = a.Cout;
  const int nct = Co / W4_CO;
  const int S = C / (4 * Q::QPP * KS);     // stages of 8 (GEO 0) / 16 (GEO 1, 2) channels of an item (KS: its half of Cin)

  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const unsigned long long uaddr = reinterpret_cast<unsigned long long>(a.w);
  const u32x4 rxv = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, (unsigned)((size_t)a.N * a.H * a.W * C * 4),
                     0x00020000u};
  const u32x4 ruv = {(unsigned)uaddr, (unsigned)(uaddr >> 32) & 0xffffu, (unsigned)((size_t)nct * (C >> 2) * W4_UKG * 4),
                     0x00020000u};
  const unsigned out_bytes = (unsigned)((size_t)a.N * a.Ho * a.Wo * Co * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, out_bytes, 0x00020000);

  // ---- halo loads: this wave's pieces wave, wave + 12 (, wave + 24); element e -> (pixel = e / QPP, channel quad e % QPP)
  int hyx[Q::NP];
  unsigned hrel[Q::NP], hws[Q::NP], hws2[Q::NP];
#pragma unroll
  for (int k = 0; k < Q::NP; ++k) {
    // LOAD order pixel-major, the channel quads of a pixel in neighbouring lanes (32 / 64 contiguous bytes: 32 or
    // 16 cache lines per instruction instead of 64); the lane stores its 16 bytes to the pixel's slots of the
    // bank-conflict-free order above
    const int e = (wave + W4_NW * k) * 64 + lane;
    const int px = e / Q::QPP, hq = e % Q::QPP;
    const int hya = px / Q::RWP, hx = px - hya * Q::RWP;        // (rows padded to whole store groups)
    const int img = hya / Q::RH, hy = hya - img * Q::RH;        // (GEO 2: the rows of the region's images follow each other)
    const bool ok = img < Q::NIMG && hx < Q::RW;
    hyx[k] = ok ? ((img << 24) | (hy << 16) | hx) : -1;
    hrel[k] = ok ? (unsigned)((((img * a.H + hy) * a.W + hx) * C + 4 * hq) * 4) : 0u;
    // (lanes past the last pixel park their zeros in the unused tail of the buffer)
    // channels 4 hq, 4 hq + 1 -> pair plane 2 hq, channels 4 hq + 2, + 3 -> pair plane 2 hq + 1
    const int slot = 2 * hq * Q::PAIR + (hx & 3) * Q::PLANE + Q::imgbase(img) + hy * Q::XD + (hx >> 2);
    hws[k] = lds0 + (unsigned)(W4_H0 + (ok ? slot * 8 : Q::HSLOT * 8 + lane * 8));
    hws2[k] = lds0 + (unsigned)(W4_H0 + (ok ? (slot + Q::PAIR) * 8 : Q::HSLOT * 8 + 512 + lane * 8));
  }
  // ---- transform share of this wave: lane (tile li of its m-tile, channel 4 g + kq of the stage)
  unsigned hb0, vw0;
  {
    const int g = GEO ? tw : (tw & 1);
    hb0 = lds0 + (unsigned)(W4_H0 + ((2 * g + (kq >> 1)) * Q::PAIR + Q::tileslot(li, tw)) * 8 + (kq & 1) * 4);
    vw0 = lds0 + (unsigned)(W4_V0 + tw * 256 + lane * 4);        // V[pt][mt * 2 + g | g][lane]
    if constexpr ((ABL & 32) != 0) hb0 = lds0 + (unsigned)(W4_H0 + lane * 4);     // conflict-free reads (wrong data)
  }
  // ---- multiply: A operands V[3 wave + pl][.][lane], filter block of this wave
  const float* va0 = smf + (W4_V0 / 4) + (3 * wave) * 256 + lane;
  const unsigned uvo = (unsigned)lane * 16u;
  // ---- exchange + output: this lane finishes tile 4 (wave & 3) + (lane >> 4) of the m-tile, co 16 (wave >> 2) + li
  const int ont = wave >> 2, okq = wave & 3;
  // exchange [point][co sub-tile][writer lane] float4 (the C fragment: tiles 4 kq_w .. + 3 of channel li_w).  The reader
  // takes ONE dword of 16 writer slots (li = 0..15): a stride of 4 dwords puts li and li + 8 on one bank (2-way conflict
  // on all 36 reads of a round).  So writers with li >= 8 store their float4 rotated by two dwords (two ds_write_b64
  // with swapped offsets -- no extra instruction), and the reader looks at dword (kq + 2 (li >> 3)) & 3.
  const unsigned xhi = (unsigned)(li >> 3);
  const unsigned xw0 = lds0 + (unsigned)((3 * wave) * 3 * 1024 + lane * 16);
  const unsigned xwa = xw0 + 8u * xhi, xwb = xw0 + 8u - 8u * xhi;           // elements (0, 1) / (2, 3) of the fragment
  const unsigned xr0 = lds0 + (unsigned)((ont * 64 + okq * 16 + li) * 16) + (((unsigned)kq + 2u * xhi) & 3u) * 4u;

  const int regs_x = a.tiles_x, regs_xy = a.tiles_x * a.tiles_y;
  const int nreg = regs_xy * ((a.N + Q::NIMG - 1) / Q::NIMG);
  const int imode = w4_item_mode(nct);     // what the XCD owns: regions / co-tiles (w4_item_mode)
  const int nwork = w4_item_count(imode, nreg, nct, KS);
  const int gsz = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  const float act_lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();
  // K split: with a ticket word per item pair (ConvArgs::tickets, programs) the second half to finish does the
  // epilogue; without one both halves add into a zeroed y (see the launcher)
  const bool tk = KS > 1 && a.tickets != nullptr;
  const bool has_res = (ABL & 64) || (KS > 1 && !tk) ? false : a.res != nullptr;
  const unsigned rowpitch = (unsigned)(a.Wo * Co) * 4u, colpitch = (unsigned)Co * 4u;

  // ABL & 64 (tools/wino4_clk.py): s_memtime stamps of every wave into the LDS above the stage buffers, dumped into
  // `res` at the end ([block][1 + 12 x 96] u64)
  constexpr int W4_NTK = 96;
  unsigned long long* sT = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(w4_smem) + w4_lds_bytes<GEO>());
  int ntk = 0;
#define W4_CLK()                                                                        \
  {                                                                                     \
    if constexpr ((ABL & 64) != 0) {                                                    \
      if (lane == 0 && ntk < W4_NTK) sT[wave * W4_NTK + ntk] = __builtin_readcyclecounter(); \
      ++ntk;                                                                            \
    }                                                                                   \
  }
  W4_CLK()
  // ST: per-(wave, statistic, channel li) sums in doubles above the stage buffers (the stamp area of the ABL & 64 build);
  // slot (wave, ., li) is written by the wave's lanes kq == 0 only; zeroed here, read after the last item (barriers between)
  double* sS = reinterpret_cast<double*>(reinterpret_cast<char*>(w4_smem) + w4_lds_bytes<GEO>());
  int ct_blk = 0;
  if constexpr (ST) {
    static_assert(ABL == 0, "the statistics build has no ablations");
    for (int e = tid; e < W4_NW * 2 * 16; e += W4_NTH) sS[e] = 0.0;
  }
  for (int w = blockIdx.x; w < nwork; w += gsz) {
    // item -> (region, co-tile): blocks w, w + 8, ... stay on one XCD (conv_wino.hip: wino8_grid)
    const unsigned wi = (unsigned)__builtin_amdgcn_readfirstlane(w);
    const unsigned xq = wi & 7u, q_ = wi >> 3;
    // (mg_nct: the multiplier of nct * KS (mode 0) or of nct / 8 (mode 2); KS is a power of two)
    const unsigned qq = w4_udiv(q_, a.mg_nct);
    int reg, ct, ks;
    if (imode == 0) {
      const int cs_ = (int)(q_ - qq * (unsigned)(nct * KS));
      reg = (int)(qq * 8u + xq); ct = cs_ / KS; ks = cs_ % KS;
    } else if (imode == 1) {
      const unsigned lg = (unsigned)nct >> 1;          // log2 of 2 / 4
      const int cs_ = (int)(q_ * (8u >> lg) + (xq >> lg));
      reg = cs_ / KS; ct = (int)(xq & ((unsigned)nct - 1u)); ks = cs_ % KS;
    } else {
      reg = (int)qq / KS; ct = (int)((q_ - qq * ((unsigned)nct >> 3)) * 8u + xq); ks = (int)qq % KS;
    }
    if (reg >= nreg) continue;                         // (uniform) padding of the last round of the XCDs
    ct_blk = ct;
    const unsigned n_ = w4_udiv((unsigned)reg, a.mg_txy);
    const unsigned r_ = (unsigned)reg - n_ * (unsigned)regs_xy;
    const unsigned ry_ = w4_udiv(r_, a.mg_tx);
    // (GEO 2: the region is the four images n .. n + 3; images past N load zeros and store nothing)
    const int n = (int)n_ * Q::NIMG, y0 = (int)ry_ * Q::RGH, x0 = (int)(r_ - ry_ * (unsigned)regs_x) * Q::RGW;

    // halo offsets of the item (stage 0): uniform base + per-lane relative offset, zero padding by OOB offsets
    unsigned doff[Q::NP];
    {
      // (KS: the item's half of the input channels starts ks * S stages into the pixel)
      const int base = ((n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * C * 4 + ks * S * Q::SBYTES;
#pragma unroll
      for (int k = 0; k < Q::NP; ++k) {
        const unsigned iy = (unsigned)(y0 - 1 + ((hyx[k] >> 16) & 0xff)), ix = (unsigned)(x0 - 1 + (hyx[k] & 0xffff));
        const bool in = hyx[k] >= 0 && iy < (unsigned)a.H && ix < (unsigned)a.W && n + (hyx[k] >> 24) < a.N;
        doff[k] = in ? (unsigned)base + hrel[k] : EGN_OOB;
      }
    }
  // The halo goes through REGISTERS (buffer_load_dwordx4 -> ds_write_b64), not LDS-DMA, and every vector-memory
  // wait of the kernel is vmcnt(0): see the note in front of W4_STAGE.
#define W4_HLOAD(K, STAGE) /* piece K of this wave; STAGE: byte offset of the stage's channels, W4_PAST = none */ \
  if constexpr ((ABL & 16) == 0) hreg[K] = w4_gld4<0>(rxv, doff[K], (unsigned)(STAGE));                        \
  else hreg[K] = f32x4{1.f, 2.f, 3.f, (float)lane};
#define W4_HLOADS(STAGE) W4_HLOAD(0, STAGE) W4_HLOAD(1, STAGE) if constexpr (Q::NP == 3) { W4_HLOAD(Q::NP - 1, STAGE) }
#define W4_HSTORE(P)                                                                                           \
  {                                                                                                            \
    _Pragma("unroll") for (int k_ = 0; k_ < Q::NP; ++k_) {                                                     \
      w4_xwr2<(P)*Q::HBYTES>(hws[k_], hreg[k_][0], hreg[k_][1]);                                               \
      w4_xwr2<(P)*Q::HBYTES>(hws2[k_], hreg[k_][2], hreg[k_][3]);                                              \
    }                                                                                                          \
  }
    // filter of this wave: k-group h (4 channels) = [co-tile][h][wave][3 x dwordx4 per lane] (engine.pack_wino4_weight);
    // a wait group is one k-group (GEO 0: x 2 m-tiles) or two consecutive ones (GEO 1).  Raw ISA -- the waits are
    // mine (tools/check_wino4_isa.py checks that no load's destination is touched before its wait)
    const unsigned ubase = (unsigned)(ct * (C >> 2) + ks * (C >> 2) / KS) * (W4_UKG * 4u) + (unsigned)wave * (3u * 64u * 16u);
#define W4_LOADB(DST, HS)                                                                                      \
  if constexpr ((ABL & 8) != 0) {                                                                              \
    _Pragma("unroll") for (int k_ = 0; k_ < Q::NKK; ++k_)                                                      \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) DST[k_][p_] = f32x4{(float)lane, 1.f, 2.f, (float)p_};    \
  } else {                                                                                                     \
    const unsigned so_ = (HS);        /* byte offset of the wait group, W4_PAST = none */                      \
    DST[0][0] = w4_gld4<0>(ruv, uvo, so_); DST[0][1] = w4_gld4<1024>(ruv, uvo, so_); DST[0][2] = w4_gld4<2048>(ruv, uvo, so_); \
    if constexpr (Q::NKK == 2) {                                                                               \
      const unsigned so1_ = so_ + W4_UKG * 4u;                                                                 \
      DST[Q::NKK - 1][0] = w4_gld4<0>(ruv, uvo, so1_); DST[Q::NKK - 1][1] = w4_gld4<1024>(ruv, uvo, so1_);     \
      DST[Q::NKK - 1][2] = w4_gld4<2048>(ruv, uvo, so1_);                                                      \
    }                                                                                                          \
  }
    constexpr unsigned WGB = (unsigned)Q::NKK * W4_UKG * 4u;       // filter bytes between consecutive wait groups
    f32x4 b0[Q::NKK][3], b1[Q::NKK][3];       // value p = 3 pl + nt of a k-group = b[kk][p >> 2][p & 3]
    f32x4 hreg[Q::NP];
    W4_HLOADS(0u)
    W4_LOADB(b0, ubase)
    W4_CLK()      /* item top: halo + filter loads issued */
    w4_vm_landedH(hreg);                                // stage 0's pieces (and the first filter wait group)
    W4_HSTORE(0)
    W4_HLOADS((unsigned)Q::SBYTES)                      // stage 1's pieces fly during the first transform
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_CLK()      /* own pieces of stage 0 in LDS */
    __builtin_amdgcn_s_barrier();
    W4_CLK()      /* everyone's */
    asm volatile("" ::: "memory");
    if constexpr ((ABL & 1) == 0) {
      if (tpart == 0) w4_transform<0, 0, GEO>(hb0, vw0);
      else if (tpart == 1) w4_transform<0, 1, GEO>(hb0, vw0);
      else w4_transform<0, 2, GEO>(hb0, vw0);
    }
    asm volatile("" ::: "memory");
    // this wave's pieces of stage 1 (and the filter loads before them) have landed: into LDS with them -- the
    // first third transforms stage 1 right behind the barrier
    w4_vm_landedH(hreg);
    W4_HSTORE(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_CLK()      /* stage 0 transformed */
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    W4_CLK()      /* K loop starts */

    f32x4 acc[3][3][Q::NMT];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int mt = 0; mt < Q::NMT; ++mt) acc[pl][nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 18 MFMAs of a wait group in three groups of 6 (one frequency point each); H0 / H1 / H2: the vector-memory
  // instruction issued behind each group -- spread over the stage instead of a burst behind the barrier, where
  // all 12 waves of the CU queue for the address unit (profiles/r3_wino4_timeline_v1.txt: 1 000 cycles per wave).
  // A operands of point pl: GEO 0 the two m-tiles of k-group G, GEO 1 the k-groups 2 G, 2 G + 1 of the one m-tile.
#define W4_MUL(P, G, B, H0, H1, H2)                                                                            \
  if constexpr ((ABL & 2) == 0) {                                                                              \
    float av_[3][2];                                                                                           \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) _Pragma("unroll") for (int x_ = 0; x_ < 2; ++x_)          \
        av_[pl][x_] = va0[((P) ? W4_VBYTES / 4 : 0) + (GEO ? (pl * 4 + 2 * (G) + x_) : ((pl * 2 + x_) * 2 + (G))) * 64]; \
    W4_MUL6(0, B) __builtin_amdgcn_sched_barrier(0); H0 __builtin_amdgcn_sched_barrier(0);                     \
    W4_MUL6(1, B) __builtin_amdgcn_sched_barrier(0); H1 __builtin_amdgcn_sched_barrier(0);                     \
    W4_MUL6(2, B) __builtin_amdgcn_sched_barrier(0); H2 __builtin_amdgcn_sched_barrier(0);                     \
  } else {                                                                                                     \
    H0 H1 H2                                                                                                   \
  }
  // (GEO 1 adds two k-groups into one accumulator: k-group outermost, so that three other MFMAs lie between them)
#define W4_MUL6(PL, B)                                                                                         \
  if constexpr (GEO == 0) {                                                                                    \
    _Pragma("unroll") for (int nt = 0; nt < 3; ++nt) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)          \
        acc[PL][nt][mt % Q::NMT] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[PL][mt], B[0][((PL)*3 + nt) >> 2][((PL)*3 + nt) & 3], \
                                                               acc[PL][nt][mt % Q::NMT], 0, 0, 0);             \
  } else {                                                                                                     \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int nt = 0; nt < 3; ++nt)          \
        acc[PL][nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[PL][kk], B[kk % Q::NKK][((PL)*3 + nt) >> 2][((PL)*3 + nt) & 3], \
                                                              acc[PL][nt][0], 0, 0, 0);                        \
  }
  // Every vector-memory wait in this kernel is s_waitcnt vmcnt(0) -- on purpose.  The first versions counted
  // (`vmcnt(3)`: "the halo pieces have landed, the three newer filter loads may stay in flight"), which needs loads
  // to return in issue order.  Measured (tools/f43_bisect.py, profiles/r3_wino4_f43_bisect.txt): alone on the GPU
  // the kernel was exact; beside other streams' kernels whole-network outputs were off by up to 10 -- and ONLY the
  // wait in which older SLOW loads (the gathered halo, HBM) sit in front of newer FAST ones (the filter, L2) had
  // to become vmcnt(0) to make it exact again, with LDS-DMA pieces and with plain register loads alike.  So the
  // stage is ordered such that no wait needs a count: wait group 2s+1 is issued at the top and awaited (alone in the
  // queue) behind the G = 0 multiplies; wait group 2s+2 is issued right there, the halo pieces of stage s + 2 behind
  // the first MFMA groups of G = 1, and all of them are awaited together at the end of the stage.
  // The halo goes through registers (buffer_load_dwordx4 -> ds_write_b64): 2 pieces per wave and stage.
  // Straight-line code: past the last stage the loads are still issued, beyond the buffers' ends (zeros).  The
  // registers are written asynchronously behind the compiler's back: the destination of a load must reach its
  // s_waitcnt without being copied -- tools/check_wino4_isa.py asserts that on the compiled ISA
  // (tests/test_wino4_design_cpu.py).
  // raised priority: at equal priority the SIMD's arbiter hands the transforming wave one VALU issue per MFMA of
  // the two multiplying waves -- 48 instructions took 1 500-2 000 cycles (profiles/r3_wino4_timeline_v2.txt)
#define W4_TRANS(P, PART)                                                                                      \
  if ((ABL & 1) == 0 && s_ + 1 < S && tpart == (PART)) {                                                       \
    __builtin_amdgcn_s_setprio(3);                                                                             \
    w4_transform<1 - (P), PART, GEO>(hb0, vw0 + (unsigned)((1 - (P)) * W4_VBYTES));                            \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  }
  // the transform of stage s + 1 sits at a different point of the stage for each third of the waves: two of a
  // SIMD's three waves always have MFMAs to issue
#define W4_STAGE(P, SI)                                                                                        \
  {                                                                                                            \
    const int s_ = (SI);                                                                                       \
    const unsigned dst_ = s_ + 2 < S ? (unsigned)(s_ + 2) * (unsigned)Q::SBYTES : W4_PAST;                     \
    const unsigned bn_ = s_ + 1 < S ? ubase + (unsigned)(2 * s_ + 2) * WGB : W4_PAST;                          \
    w4_vm_landedB(b0);                                                                                         \
    W4_CLK() /* 0: filter wait group 2s landed */                                                              \
    W4_LOADB(b1, s_ < S ? ubase + (unsigned)(2 * s_ + 1) * WGB : W4_PAST)                                      \
    W4_TRANS(P, 0)                                                                                             \
    W4_CLK() /* 1: loads issued (+ transform, first third of the waves) */                                     \
    W4_MUL(P, 0, b0, , , )                                                                                     \
    W4_CLK() /* 2: G = 0 multiplies issued */                                                                  \
    w4_vm_landedB(b1);                                                                                         \
    W4_CLK() /* 3: filter wait group 2s+1 landed */                                                            \
    W4_LOADB(b0, bn_)       /* (its registers are free: G = 0 is issued) a whole half stage of flight */       \
    W4_TRANS(P, 1)                                                                                             \
    W4_MUL(P, 1, b1, W4_HLOAD(0, dst_), W4_HLOAD(1, dst_), if constexpr (Q::NP == 3) { W4_HLOAD(Q::NP - 1, dst_) })   \
    W4_TRANS(P, 2)                                                                                             \
    W4_CLK() /* 4: G = 1 multiplies and wait group 2s+2 issued (+ transforms) */                               \
    w4_vm_landedH(hreg);             /* the pieces of stage s + 2 and wait group 2s+2 */                       \
    W4_HSTORE(P)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    W4_CLK() /* 5: own pieces of stage s + 2 in LDS, V writes done */                                          \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("" ::: "memory");                                                                             \
    W4_CLK() /* 6: past the barrier */                                                                         \
  }
    for (int s = 0; s + 1 < S; s += 2) {     // (GEO 0: S is even, Cin % 16 == 0)
      W4_STAGE(0, s)
      W4_STAGE(1, s + 1)
    }
    if constexpr (GEO != 0) {
      if (S & 1) W4_STAGE(0, S - 1)          // 48 / 16 = 3 stages: the odd one (every item starts at parity 0)
    }
    // the loads past the end: tied to the wait -- for the compiler their registers are dead at the loop exit, and it
    // may move arithmetic of the item end into them while the loads are still in flight (seen with another form of
    // the item end, caught by tools/check_wino4_isa.py)
    w4_vm_landedB(b0);
    w4_vm_landedB(b1);
    w4_vm_landedH(hreg);
    W4_CLK()      /* K loop done */
#undef W4_STAGE
#undef W4_TRANS
#undef W4_MUL
#undef W4_MUL6
#undef W4_HLOAD
#undef W4_HLOADS
#undef W4_HSTORE
#undef W4_LOADB

    // ---- item end: per m-tile, accumulators -> LDS -> one (tile, co) per lane -> Y = A^T M A -> epilogue
    const float sc = a.scale[ct * W4_CO + ont * 16 + li];
    const float sh = a.shift[ct * W4_CO + ont * 16 + li];
    if constexpr ((ABL & 4) != 0) {
      f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
          for (int mt = 0; mt < Q::NMT; ++mt) t += acc[pl][nt][mt];
      if (t[0] + t[1] + t[2] + t[3] == 12345.f) a.y[tid] = t[0];
      continue;
    }
#pragma unroll
    for (int mt = 0; mt < Q::NMT; ++mt) {
      // this lane's output tile of the round: tile 4 okq + kq of m-tile mt, channel 16 ont + li
      const int tile = 4 * okq + kq;
      const int oimg = GEO == 2 ? (tile >> 2) : 0;
      const int ty = GEO == 0 ? (2 * mt + (tile >> 3)) : (GEO == 1 ? (tile >> 2) : ((tile >> 1) & 1));
      const int tx = GEO == 0 ? (tile & 7) : (GEO == 1 ? (tile & 3) : (tile & 1));
      const unsigned vo = n + oimg < a.N ? (unsigned)(((((n + oimg) * a.Ho + y0 + 4 * ty) * a.Wo + x0 + 4 * tx) * Co +
                                                       ct * W4_CO + ont * 16 + li) * 4)
                                         : EGN_OOB;
      float rv[4][4];
#pragma unroll
      for (int oa = 0; oa < 4; ++oa)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
          rv[oa][ob] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     rr, has_res ? vo : EGN_OOB, oa * rowpitch + ob * colpitch, 0));
#define W4_XW(SL, V) w4_xwr2<(SL)*1024>(xwa, (V)[0], (V)[1]); w4_xwr2<(SL)*1024>(xwb, (V)[2], (V)[3]);
      W4_XW(0, acc[0][0][mt]) W4_XW(1, acc[0][1][mt]) W4_XW(2, acc[0][2][mt])
      W4_XW(3, acc[1][0][mt]) W4_XW(4, acc[1][1][mt]) W4_XW(5, acc[1][2][mt])
      W4_XW(6, acc[2][0][mt]) W4_XW(7, acc[2][1][mt]) W4_XW(8, acc[2][2][mt])
#undef W4_XW
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      W4_CLK()    /* round: accumulators written */
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      W4_CLK()    /* round: exchange barrier passed */
      // M[i][j] = X[6 i + j]: two columns per batch (12 reads), column pass A^T right away: 24 values stay
      float yc[4][6];
#define W4_M(I, J) w4_xrd<(I)*6 + (J)>(xr0, xr1)
#define W4_COLS(J0)                                                                                     \
  {                                                                                                     \
    float ca_[6], cb_[6], ya_[4], yb_[4];                                                               \
    ca_[0] = W4_M(0, J0); ca_[1] = W4_M(1, J0); ca_[2] = W4_M(2, J0); ca_[3] = W4_M(3, J0); ca_[4] = W4_M(4, J0);  \
    ca_[5] = W4_M(5, J0);                                                                               \
    cb_[0] = W4_M(0, J0 + 1); cb_[1] = W4_M(1, J0 + 1); cb_[2] = W4_M(2, J0 + 1); cb_[3] = W4_M(3, J0 + 1);        \
    cb_[4] = W4_M(4, J0 + 1); cb_[5] = W4_M(5, J0 + 1);                                                 \
    w4_landed6(ca_, cb_);                                                                               \
    w4_at(ca_, ya_);                                                                                    \
    w4_at(cb_, yb_);                                                                                    \
    _Pragma("unroll") for (int oa = 0; oa < 4; ++oa) { yc[oa][J0] = ya_[oa]; yc[oa][J0 + 1] = yb_[oa]; } \
  }
      const unsigned xr1 = xr0 + 18u * 3072u;
      W4_COLS(0)
      W4_COLS(2)
      W4_COLS(4)
#undef W4_COLS
#undef W4_M
      double st1 = 0.0, st2 = 0.0;       // ST: this lane's 16 stored values of the round (one channel), in doubles from the
                                         // first add (a mean far above the deviation loses the variance's digits in fp32: ADVICE r5)
#pragma unroll
      for (int oa = 0; oa < 4; ++oa) {
        float yo[4];
        w4_at(yc[oa], yo);
        if constexpr (KS > 1) {
#pragma unroll
          for (int ob = 0; ob < 4; ++ob) yc[oa][ob] = yo[ob];      // (the column values of this row are spent)
        } else {
#pragma unroll
          for (int ob = 0; ob < 4; ++ob) {
            const float v = fmaxf(__builtin_fmaf(yo[ob], sc, sh) + rv[oa][ob], act_lo);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, vo, oa * rowpitch + ob * colpitch, 0);
            if constexpr (ST) { st1 += (double)v; st2 = __builtin_fma((double)v, (double)v, st2); }
          }
        }
      }
      if constexpr (KS > 1) {
        // yc[oa][ob]: this item's share of the output sum (its half of the input channels)
        if (!tk) {      // no ticket word: add into the zeroed y; conv_wino4_finish_kernel does the rest
#pragma unroll
          for (int oa = 0; oa < 4; ++oa)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
              __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(yc[oa][ob], ry, vo, oa * rowpitch + ob * colpitch, 0);
        } else {
          // The two halves of an item pair meet through a ticket word (zero between launches).  Whoever finishes its
          // K loop first writes its raw share THROUGH to memory (sc1 stores into y: no release fence -- a buffer_wbl2
          // per block costs microseconds and all 768 lanes fencing 4x that), drains its stores and bumps the word to
          // 3; the other half polls for that with relaxed loads (its partner is already in its item end: no dependence
          // on dispatch order or placement), reads the share back with sc1 loads (past the CU's L1: no acquire fence,
          // MI355X_MICROARCH.md "inter-workgroup visibility": sc1 stores AND sc1 loads), adds its own and applies the
          // epilogue.  Two addends: the sum does not depend on who came first.
          constexpr int SC1 = 16;            // aux bit of the raw buffer builtins
          unsigned* tkw = a.tickets + 1 + (reg * nct + ct);      // (word 0: the error word of whoever owns the words)
          unsigned* tks = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(w4_smem) + W4_XBYTES);   // (past the exchange)
          if (tid == 0) *tks = __hip_atomic_fetch_add(tkw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __syncthreads();
          const bool first = __builtin_amdgcn_readfirstlane(*tks) == 0u;
          if (first) {
#pragma unroll
            for (int oa = 0; oa < 4; ++oa)
#pragma unroll
              for (int ob = 0; ob < 4; ++ob)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yc[oa][ob]), ry, vo,
                                                      oa * rowpitch + ob * colpitch, SC1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its stores have left the CU
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(tkw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            if (tid == 0) {
              // bounded [round 5, ADVICE r4]: a word that is not zero at launch (a program run beside itself, a caller
              // that shares words between streams) would make both halves wait for ever; after ~2^22 polls (seconds)
              // the block gives up, raises the owner's error word -- word 0 of the buffer, the same index for every layer
              // that shares the words (the training tape) [round 6, ADVICE r5]: programs read it back behind every run
              // and fail the NEXT run (csrc/program.hip), the tape's owner checks it where it reads the loss -- and goes on
              // with whatever y holds
              unsigned spins = 0;
              while (__hip_atomic_load(tkw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 3u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) {
                  __hip_atomic_store(a.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  break;
                }
              }
            }
            __syncthreads();
            float pv[4][4];
#pragma unroll
            for (int oa = 0; oa < 4; ++oa)
#pragma unroll
              for (int ob = 0; ob < 4; ++ob)
                pv[oa][ob] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, vo, oa * rowpitch + ob * colpitch, SC1));
#pragma unroll
            for (int oa = 0; oa < 4; ++oa)
#pragma unroll
              for (int ob = 0; ob < 4; ++ob) {
                const float v = fmaxf(__builtin_fmaf(yc[oa][ob] + pv[oa][ob], sc, sh) + rv[oa][ob], act_lo);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, vo, oa * rowpitch + ob * colpitch, 0);
                if constexpr (ST) { st1 += (double)v; st2 = __builtin_fma((double)v, (double)v, st2); }
              }
            if (tid == 0) __hip_atomic_store(tkw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      if constexpr (ST) {
        // 16 values per lane, the four tiles of a channel (kq) by shuffles, then into the wave's doubles.  With the K
        // split the half that finishes SECOND stores the outputs and owns their statistics: which block's row gets them
        // depends on arrival order, the total over the rows (egn_bn_stats_finalize_f32 adds them in a fixed order of
        // rows, each a sum of doubles) only in the last bits of a double -- far below the fp32 mean / variance it feeds.
        if (vo == EGN_OOB) { st1 = 0.0; st2 = 0.0; }          // (GEO 2: an image past N stores nothing)
        st1 += __shfl_xor(st1, 16); st1 += __shfl_xor(st1, 32);
        st2 += __shfl_xor(st2, 16); st2 += __shfl_xor(st2, 32);
        if (kq == 0) {
          sS[(wave * 2 + 0) * 16 + li] += st1;
          sS[(wave * 2 + 1) * 16 + li] += st2;
        }
      }
      asm volatile("" ::: "memory");
      W4_CLK()    /* round: output transform done, stores issued */
      __builtin_amdgcn_s_barrier();      // the exchange buffer is free again (next round / next item's DMA)
      asm volatile("" ::: "memory");
      W4_CLK()    /* round: end */
    }
  }
  if constexpr (ST) {
    // the block's row of the partial table: [2][Cout] doubles, its co-tile's 48 channels, zeros elsewhere
    __syncthreads();
    double* row = a.stats + (size_t)blockIdx.x * 2 * Co;
    for (int e = tid; e < 2 * Co; e += W4_NTH) {
      const int which = e / Co, c = e - which * Co;
      const int cl = c - ct_blk * W4_CO;
      double v = 0.0;
      if (cl >= 0 && cl < W4_CO) {
        const int nt_ = cl >> 4, l_ = cl & 15;       // the lanes of waves 4 nt .. 4 nt + 3 store co sub-tile nt
#pragma unroll
        for (int k = 0; k < 4; ++k) v += sS[((4 * nt_ + k) * 2 + which) * 16 + l_];
      }
      row[e] = v;
    }
  }
  if constexpr ((ABL & 64) != 0) {
    __syncthreads();
    unsigned long long* out = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res)) +
                              (size_t)blockIdx.x * (W4_NW * W4_NTK + 1);
    for (int e = tid; e < W4_NW * W4_NTK; e += W4_NTH) out[1 + e] = sT[e];
    if (tid == 0) out[0] = (unsigned long long)ntk;
  }
#undef W4_CLK
}

template <int ABL>
__global__ __launch_bounds__(W4_NTH, 1) void conv_wino4_kernel(ConvArgs a) { w4_body<ABL, 0>(a); }
template <int ABL>
__global__ __launch_bounds__(W4_NTH, 1) void conv_wino4b_kernel(ConvArgs a) { w4_body<ABL, 1>(a); }
template <int ABL>      // conv_wino4b_kernel with the input channels of an item split over two blocks (as conv_wino4c_kernel<., 2>)
__global__ __launch_bounds__(W4_NTH, 1) void conv_wino4bk_kernel(ConvArgs a) { w4_body<ABL, 1, 2>(a); }
template <int ABL, int KS>
__global__ __launch_bounds__(W4_NTH, 1) void conv_wino4c_kernel(ConvArgs a) { w4_body<ABL, 2, KS>(a); }
// the training tape's builds (BatchNorm statistics in the item end, `ConvArgs::stats`): every geometry / K split above
template <int GEO, int KS>
__global__ __launch_bounds__(W4_NTH, 1) void conv_wino4s_kernel(ConvArgs a) { w4_body<0, GEO, KS, true>(a); }

// second pass of the K-split form: y holds the sum of the items' raw outputs; y = act(y * scale + shift + res) in place
__global__ __launch_bounds__(256) void conv_wino4_finish_kernel(float* __restrict__ y, const float* __restrict__ res,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, unsigned n4, unsigned co4,
                                                                float act_lo) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n4) return;
  const unsigned c = i % co4;
  const float4 sc = reinterpret_cast<const float4*>(scale)[c], sh = reinterpret_cast<const float4*>(shift)[c];
  float4 v = reinterpret_cast<float4*>(y)[i];
  const float4 r = res ? reinterpret_cast<const float4*>(res)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  v.x = fmaxf(__builtin_fmaf(v.x, sc.x, sh.x) + r.x, act_lo);
  v.y = fmaxf(__builtin_fmaf(v.y, sc.y, sh.y) + r.y, act_lo);
  v.z = fmaxf(__builtin_fmaf(v.z, sc.z, sh.z) + r.z, act_lo);
  v.w = fmaxf(__builtin_fmaf(v.w, sc.w, sh.w) + r.w, act_lo);
  reinterpret_cast<float4*>(y)[i] = v;
}

// geo: bits 0-1 the geometry (0: 16 x 32 regions, 1: 16 x 16 regions, 2: four 8 x 8 images), bit 2: input channels split
// over two items (geometries 1 and 2: one exchange round per item)
// bit 3 (geo 9) [round 6]: 96 output channels per item on geometry 1 -- conv_wino4w.hip
bool egn_conv_wino4w_applies(const ConvArgs& a);
int egn_conv_launch_wino4w(ConvArgs a, size_t lds, int abl, hipStream_t stream);
// bit 4 (geo 17) [round 6]: half-size blocks (6 waves, 16 tiles), two per CU -- conv_wino4h.hip
bool egn_conv_wino4h_applies(const ConvArgs& a);
// bit 5 (geo 49): the two blocks of a CU as the halves of ONE 12-wave workgroup (conv_wino4d_kernel)
size_t egn_conv_wino4h_lds_bytes(int dual);
int egn_conv_launch_wino4h(ConvArgs a, size_t lds, int abl, int dual, hipStream_t stream);
// bit 6 (geo 64) [round 6]: row-owner waves, one exchange round per item, 16 x 32 regions -- conv_wino4r.hip
bool egn_conv_wino4r_applies(const ConvArgs& a);
size_t egn_conv_wino4r_lds_bytes();
int egn_conv_launch_wino4r(ConvArgs a, size_t lds, int abl, hipStream_t stream);
bool egn_conv_wino4_applies(const ConvArgs& a, int geo) {
  if (geo & 64) return geo == 64 && egn_conv_wino4r_applies(a);
  if (geo & 16) return (geo == 17 || geo == 49) && egn_conv_wino4h_applies(a);
  if (geo & 8) return geo == 9 && egn_conv_wino4w_applies(a);
  const int g = geo & 3, ks = (geo & 4) ? 2 : 1;
  if (g > 2 || (ks > 1 && g == 0)) return false;
  const bool map_ok = g == 2 ? (a.Ho == 8 && a.Wo == 8) : (a.Ho % 16 == 0 && a.Wo % (g ? 16 : 32) == 0);
  // (K split: whole 16-channel stages per half; the residual is read after y was zeroed -- it must be another buffer)
  if (ks > 1 && (a.Cin % 32 || (a.res && a.res == a.y))) return false;
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 16 == 0 && a.cs_in == a.Cin &&
         a.Cout % W4_CO == 0 && a.cs_out == a.Cout && !a.out_nchw && map_ok && !(a.act & EGN_ACT_RES_AFTER) &&
         ((a.act & EGN_ACT_MASK) == EGN_ACT_NONE || (a.act & EGN_ACT_MASK) == EGN_ACT_RELU);
}
// ticket words a K-split launch wants (zeroed once; every launch leaves them zero): the error word (index 0) + one per
// (region, co-tile); 0 = none
// (a: planned -- tiles_x / tiles_y are the regions of an image)
int egn_conv_wino4_tickets(const ConvArgs& a, int geo) {
  if (!(geo & 4) || !egn_conv_wino4_applies(a, geo)) return 0;
  const int nimg = (geo & 3) == 2 ? 4 : 1;
  // the error word (raised by a block whose wait ran out) + one word per (region, co-tile)
  return a.tiles_x * a.tiles_y * ((a.N + nimg - 1) / nimg) * (a.Cout / W4_CO) + 1;
}
size_t egn_conv_wino4_lds_bytes(int geo) {      // (+ the stamp area of the ABL & 64 build)
  if (geo & 64) return egn_conv_wino4r_lds_bytes();
  if (geo & 16) return egn_conv_wino4h_lds_bytes(geo & 32);
  return ((geo & 3) == 2 ? w4_lds_bytes<2>() : w4_lds_bytes<0>()) + 12 * 96 * 8;
}
// floats of the packed filter (engine.pack_wino4_weight): [co-tile][k-group = Cin / 4][wave][9 of 12][64]
extern "C" long long egn_wino4_weight_floats(int cout, int cin) {
  if (cout % W4_CO || cin % 8 || cin < 16) return 0;
  return (long long)(cout / W4_CO) * (cin / 8) * 2 * W4_UKG;
}

// Filter transform on the device: torch weight [Cout][Cin][3][3] -> U = G g G^T (float64 arithmetic, one rounding to fp32)
// in the register-feed layout above -- what engine.pack_wino4_weight computes on the host.  dgrad = 1: the data-gradient
// filter (in / out channels swapped, taps rotated by 180 degrees), as egn_wino_pack_weight_f32 does for F(2x2,3x3).  One
// thread per (co, ci): 9 loads, 36 stores; the thread of a co-tile's first 16 channels also zeroes the three padding
// values per (wave, lane).  (Training weights change every step: the host transform cannot feed the tape.)
__global__ __launch_bounds__(256) void wino4_pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int dgrad,
                                                                float* __restrict__ dst) {
  const int n_out = dgrad ? Cin : Cout, n_in = dgrad ? Cout : Cin;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_out * n_in; e += gridDim.x * blockDim.x) {
    const int o = e / n_in, i = e - o * n_in;
    w4p_pack_pair(w, Cin, dgrad, o, i, n_in, dst);
  }
}
extern "C" long long egn_wino4_pack_weight_floats(int Cout, int Cin, int dgrad) {
  return dgrad ? egn_wino4_weight_floats(Cin, Cout) : egn_wino4_weight_floats(Cout, Cin);
}
extern "C" int egn_wino4_pack_weight_f32(const float* w, int Cout, int Cin, int dgrad, float* dst, void* stream) {
  if (!w || !dst || egn_wino4_pack_weight_floats(Cout, Cin, dgrad) == 0) return EGN_E_BADARG;
  const long total = (long)Cout * Cin;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(wino4_pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, dgrad, dst);
  return (int)hipGetLastError();
}

static unsigned w4_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

static int w4_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}
// persistent grid: one block per CU in whole XCD rounds of (co-tile, K half) -- a multiple of 8 nct KS, so that the
// items w, w + grid, ... of a block share the co-tile (and the K half): its BatchNorm partial row covers one co-tile
static int w4_grid(int nwork, int nck) {
  int cap = w4_cus() / (8 * nck) * (8 * nck);
  if (cap <= 0) cap = 8 * nck;
  return nwork < cap ? nwork : cap;
}
// rows of the BatchNorm partial table a launch with ConvArgs::stats writes (a: planned): one per block; 0 = none
int egn_conv_wino4_stats_rows(const ConvArgs& a, int geo) {
  if ((geo & (24 | 64)) || !egn_conv_wino4_applies(a, geo)) return 0;      // (the wide items / half blocks have no training build)
  const int g = geo & 3, ks = (geo & 4) ? 2 : 1, nimg = g == 2 ? 4 : 1;
  const int nct = a.Cout / W4_CO;
  const int nreg = a.tiles_x * a.tiles_y * ((a.N + nimg - 1) / nimg);
  return w4_grid(w4_item_count(w4_item_mode(nct), nreg, nct, ks), nct * ks);
}

template <int ABL, int GEO, int KS, bool ST = false>
static int wino4_launch(ConvArgs a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];
  void (*kern)(ConvArgs);
  if constexpr (ST) kern = &conv_wino4s_kernel<GEO, KS>;
  else kern = GEO == 2 ? &conv_wino4c_kernel<ABL, KS> : (GEO == 0 ? &conv_wino4_kernel<ABL> : (KS > 1 ? &conv_wino4bk_kernel<ABL> : &conv_wino4b_kernel<ABL>));
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 512));
  }
  if (ST && KS > 1 && !a.tickets) return EGN_E_BADARG;   // (the three-launch form applies the epilogue in another kernel)
  const int nct = a.Cout / W4_CO, nck = nct * KS;
  const int nreg = a.tiles_x * a.tiles_y * ((a.N + W4G<GEO>::NIMG - 1) / W4G<GEO>::NIMG);
  const int imode = w4_item_mode(nct);
  const int nwork = w4_item_count(imode, nreg, nct, KS);
  // the item index divisions run as one multiply-high each: exact while x * d < 2^32 (x = the dividend's range)
  if ((unsigned long long)nwork * (unsigned)(8 * nck) >= 0x100000000ull ||
      (unsigned long long)(nreg + 8) * (unsigned)(a.tiles_x * a.tiles_y) >= 0x100000000ull)
    return EGN_E_BADARG;
  a.mg_nct = imode == 1 ? 0u : w4_magic(imode == 2 ? nct / 8 : nck);
  a.mg_txy = w4_magic(a.tiles_x * a.tiles_y);
  a.mg_tx = w4_magic(a.tiles_x);
  const int grid = w4_grid(nwork, nck);               // one block per CU, whole XCD rounds
  if (KS > 1 && !a.tickets) {
    const size_t n = (size_t)a.N * a.Ho * a.Wo * a.Cout;
    EGN_CHECK_HIP(hipMemsetAsync(a.y, 0, n * sizeof(float), stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(W4_NTH), lds, stream, a);
    const float act_lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();
    hipLaunchKernelGGL(conv_wino4_finish_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, a.y, a.res,
                       a.scale, a.shift, (unsigned)(n / 4), (unsigned)(a.Cout / 4), act_lo);
  } else {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(W4_NTH), lds, stream, a);
  }
  return (int)hipGetLastError();
}
int egn_conv_launch_wino4(ConvArgs a, size_t lds, int abl, int geo, hipStream_t stream) {
  if (!egn_conv_wino4_applies(a, geo)) return EGN_E_BADARG;
  if (geo & 64) return a.stats ? EGN_E_BADARG : egn_conv_launch_wino4r(a, lds, abl, stream);
  if (geo & 16) return a.stats ? EGN_E_BADARG : egn_conv_launch_wino4h(a, lds, abl, geo & 32, stream);
  if (geo & 8) return a.stats ? EGN_E_BADARG : egn_conv_launch_wino4w(a, lds, abl, stream);
  if (a.stats) {                       // the training tape: BatchNorm statistics in the item end
    if (abl) return EGN_E_BADARG;
    switch (geo) {
      case 0: return wino4_launch<0, 0, 1, true>(a, lds, stream);
      case 1: return wino4_launch<0, 1, 1, true>(a, lds, stream);
      case 5: return wino4_launch<0, 1, 2, true>(a, lds, stream);
      case 2: return wino4_launch<0, 2, 1, true>(a, lds, stream);
      case 6: return wino4_launch<0, 2, 2, true>(a, lds, stream);
      default: return EGN_E_BADARG;
    }
  }
  if ((geo & 3) == 2) {
    if (abl) return EGN_E_BADARG;
    return (geo & 4) ? wino4_launch<0, 2, 2>(a, lds, stream) : wino4_launch<0, 2, 1>(a, lds, stream);
  }
  if (geo == 5) return abl ? EGN_E_BADARG : wino4_launch<0, 1, 2>(a, lds, stream);
  if (geo) {
    switch (abl) {
      case 0: return wino4_launch<0, 1, 1>(a, lds, stream);
#ifdef EGN_PROBES
      case 64: return wino4_launch<64, 1, 1>(a, lds, stream);
#endif
      default: return EGN_E_BADARG;
    }
  }
  switch (abl) {
    case 0: return wino4_launch<0, 0, 1>(a, lds, stream);
#ifdef EGN_PROBES       // timing ablations / stamp builds: tools/ only (python -m egonet_amd.build --probes)
    case 1: return wino4_launch<1, 0, 1>(a, lds, stream);
    case 2: return wino4_launch<2, 0, 1>(a, lds, stream);
    case 4: return wino4_launch<4, 0, 1>(a, lds, stream);
    case 8: return wino4_launch<8, 0, 1>(a, lds, stream);
    case 16: return wino4_launch<16, 0, 1>(a, lds, stream);
    case 7: return wino4_launch<7, 0, 1>(a, lds, stream);
    case 32: return wino4_launch<32, 0, 1>(a, lds, stream);
    case 64: return wino4_launch<64, 0, 1>(a, lds, stream);
#endif
    default: return EGN_E_BADARG;
  }
}
