// conv_wino4w.hip -- fused Winograd F(4x4,3x3), 96 output channels per item ("wide" items) [round 6].
//
// Why.  conv_wino4_kernel's K stage takes ~5 600 cycles for 3 456 cycles of MFMA issue per SIMD (profiles/
// r3_wino4_timeline_v2.txt, r5_pmc_sq_wino4.txt: MFMA busy 43 % of CU-busy).  What it pays beside the MFMAs is per
// (tile, input channel): the input transform V = B^T d B (144 VALU per 64 pairs, each VALU instruction 3-10 cycles of
// matrix-pipe time, profiles/r3_mfma_tax.txt), the halo loads and stores, the A-operand reads -- and per stage a barrier
// over 12 waves (~1 000 cycles of skew).  None of that depends on how many OUTPUT channels multiply the transformed
// tile.  With one 48-channel co-tile per item a (tile, channel) pair feeds 3 MFMAs per frequency point; here an item is
// 16 tiles x 96 output channels = TWO co-tiles of the same packed filter, so a pair feeds 6: half the transform
// instructions, halo bytes, A reads and barriers per MFMA, the same 72 accumulator registers per wave as
// conv_wino4_kernel (3 points x 6 co sub-tiles x 1 m-tile instead of 3 x 3 x 2).
//
// Block = 12 waves, region = 16 x 16 output pixels of one image (conv_wino4b_kernel's geometry 1: halo order, transform
// shares, V layout [point][k-group 0..3][lane], 16-channel stages), wave w owns frequency points 3w .. 3w+2.
// A wait group is ONE 4-channel k-group: 6 buffer_load_dwordx4 per lane (the wave's slices of co-tiles 2p and 2p + 1 in
// wino4_pack.h's layout -- no new packing), 18 MFMAs; four wait groups per stage, two filter buffers in registers.
// The transform thirds sit where only ONE filter buffer is live (72 accumulators + 24 filter + 8 halo + ~40 transform
// registers: 168 is the budget of three waves per SIMD).  Every vector-memory wait is vmcnt(0) (conv_wino4.hip).
// Item end: two exchange rounds (one per co-tile) through the 108 KB [point][co sub-tile][lane] float4 image that
// conv_wino4_kernel uses per m-tile.
// Which layers: 96 -> 96 @ 32 x 32 at 64 crops (256 items; 64 launches of the W48 forward).  192 / 384 channels have
// too few items at 64 crops (128 / 64); the tuner measures, the table decides.
// Reference: the 3x3 stride-1 convolutions of libs/model/heatmapModel/hrnet.py (BasicBlock :49-76).
#include "conv_wino4.h"

namespace {
typedef W4G<1> QW;
constexpr int W4W_NT = 6;                          // co sub-tiles of an item: two 48-channel co-tiles
constexpr unsigned W4W_KGB = W4_UKG * 4u;          // filter bytes of one (co-tile, k-group)
}  // namespace

// The filter registers of a k-group: the 9 used values (p = 3 pl + nt') of the 12 per (wave, lane, co-tile) slot --
// values 0 .. 7 as two dwordx4, value 8 as one dword (18 registers per buffer instead of 24: the budget)
struct W4WB {
  f32x4 q[4];       // [2 c + (p >> 2)][p & 3], p < 8
  float s[2];       // value 8 of co-tile c
};
__device__ __forceinline__ void w4w_vm_landedB(W4WB& b) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(b.q[0]), "+v"(b.q[1]), "+v"(b.q[2]), "+v"(b.q[3]), "+v"(b.s[0]), "+v"(b.s[1]));
}
template <int OFF>
__device__ __forceinline__ float w4w_gld1(u32x4 rsrc, unsigned voff, unsigned soff) {
  float v;
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF));
  return v;
}
__device__ __forceinline__ float w4w_bval(const W4WB& b, int c, int p) { return p < 8 ? b.q[2 * c + (p >> 2)][p & 3] : b.s[c]; }

// ABL (probe builds): bit 6 s_memtime stamps of every wave (tools/wino4_clk.py), dumped into `res`.
template <int ABL>
__device__ __forceinline__ void w4w_body(const ConvArgs& a) {
  typedef QW Q;
  extern __shared__ float4 w4_smem[];
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_w4_t)w4_smem;
  const float* smf = reinterpret_cast<const float*>(w4_smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int tpart = wave >> 2;             // its third of the frequency rows of the input transform
  const int tw = wave & 3;                 // its k-group of the stage in the transform

  const int C = a.Cin, Co = a.Cout;
  const int nct = Co / W4_CO, ncp = nct >> 1;        // co-tiles, co-tile pairs
  const int S = C / 16;                    // stages of 16 channels

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

  // ---- halo loads (conv_wino4.hip, geometry 1): pieces wave, wave + 12; element e -> (pixel e / 4, channel quad e % 4).
  // Every lane stores to the NATURAL slot of its element: the padded columns 18 / 19 of a row fall on slots (x & 3) in
  // {2, 3}, x >> 2 = 4 and the rows 18 / 19 past the halo on slots 90 .. 95 of a plane -- none of them is read by the
  // transform (rows 0 .. 17 x (x >> 2) 0 .. 4 of the planes of x = 0 .. 17), so the idle lanes need no parking area and
  // the second channel pair of a lane sits at a constant + PAIR slots (one address register per piece).
  static_assert(Q::RH * Q::XD + 5 < Q::PLANE && 7 * Q::PAIR + 3 * Q::PLANE + 19 * Q::XD + 5 <= Q::HSLOT, "natural slots of the idle lanes");
  unsigned hws[Q::NP];
#pragma unroll
  for (int k = 0; k < Q::NP; ++k) {
    const int e = (wave + W4_NW * k) * 64 + lane;
    const int px = e / Q::QPP, hq = e % Q::QPP;
    const int hy = px / Q::RWP, hx = px - hy * Q::RWP;
    const int slot = 2 * hq * Q::PAIR + (hx & 3) * Q::PLANE + hy * Q::XD + (hx >> 2);
    hws[k] = lds0 + (unsigned)(W4_H0 + slot * 8);
  }
  // ---- transform share: lane (tile li, channel 4 tw + kq of the stage)
  const unsigned hb0 = lds0 + (unsigned)(W4_H0 + ((2 * tw + (kq >> 1)) * Q::PAIR + Q::tileslot(li, tw)) * 8 + (kq & 1) * 4);
  const unsigned vw0 = lds0 + (unsigned)(W4_V0 + tw * 256 + lane * 4);        // V[pt][g][lane]
  // ---- multiply: A operands V[3 wave + pl][g][lane]
  const float* va0 = smf + (W4_V0 / 4) + (3 * wave) * 256 + lane;
  const unsigned uvo = (unsigned)lane * 16u;
  const int regs_x = a.tiles_x, regs_xy = a.tiles_x * a.tiles_y;
  const int nreg = regs_xy * a.N;
  const int imode = w4_item_mode(ncp);     // what the XCD owns, over co-tile PAIRS
  const int nwork = w4_item_count(imode, nreg, ncp, 1);
  const int gsz = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  const float act_lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();
  const bool has_res = (ABL & 64) ? false : a.res != nullptr;
  const unsigned rowpitch = (unsigned)(a.Wo * Co) * 4u, colpitch = (unsigned)Co * 4u;
  const unsigned ctstride = (unsigned)(C >> 2) * W4W_KGB;        // filter bytes of a co-tile

  constexpr int W4_NTK = 96;
  unsigned long long* sT = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(w4_smem) + w4_lds_bytes<1>());
  int ntk = 0;
#define W4_CLK()                                                                        \
  {                                                                                     \
    if constexpr ((ABL & 64) != 0) {                                                    \
      if (lane == 0 && ntk < W4_NTK) sT[wave * W4_NTK + ntk] = __builtin_readcyclecounter(); \
      ++ntk;                                                                            \
    }                                                                                   \
  }
  W4_CLK()
  for (int w = blockIdx.x; w < nwork; w += gsz) {
    const unsigned wi = (unsigned)__builtin_amdgcn_readfirstlane(w);
    const unsigned xq = wi & 7u, q_ = wi >> 3;
    const unsigned qq = w4_udiv(q_, a.mg_nct);
    int reg, cp;
    if (imode == 0) {
      cp = (int)(q_ - qq * (unsigned)ncp);
      reg = (int)(qq * 8u + xq);
    } else if (imode == 1) {
      const unsigned lg = (unsigned)ncp >> 1;
      reg = (int)(q_ * (8u >> lg) + (xq >> lg)); cp = (int)(xq & ((unsigned)ncp - 1u));
    } else {
      reg = (int)qq; cp = (int)((q_ - qq * ((unsigned)ncp >> 3)) * 8u + xq);
    }
    if (reg >= nreg) continue;
    const int ct0 = 2 * cp;
    const unsigned n_ = w4_udiv((unsigned)reg, a.mg_txy);
    const unsigned r_ = (unsigned)reg - n_ * (unsigned)regs_xy;
    const unsigned ry_ = w4_udiv(r_, a.mg_tx);
    const int n = (int)n_, y0 = (int)ry_ * Q::RGH, x0 = (int)(r_ - ry_ * (unsigned)regs_x) * Q::RGW;

    // halo offsets of the item: recomputed from the lane id per item (not kept over the K loop: registers)
    unsigned doff[Q::NP];
    {
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      const int base = ((n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * C * 4;
#pragma unroll
      for (int k = 0; k < Q::NP; ++k) {
        const int e = (wave + W4_NW * k) * 64 + lane_t;
        const int px = e / Q::QPP, hq = e % Q::QPP;
        const int hy = px / Q::RWP, hx = px - hy * Q::RWP;
        const unsigned iy = (unsigned)(y0 - 1 + hy), ix = (unsigned)(x0 - 1 + hx);
        const bool in = hy < Q::RH && hx < Q::RW && iy < (unsigned)a.H && ix < (unsigned)a.W;
        doff[k] = in ? (unsigned)(base + ((hy * a.W + hx) * C + 4 * hq) * 4) : EGN_OOB;
      }
    }
#define W4_HLOAD(K, STAGE) hreg[K] = w4_gld4<0>(rxv, doff[K], (unsigned)(STAGE));
#define W4_HLOADS(STAGE) W4_HLOAD(0, STAGE) W4_HLOAD(1, STAGE)
#define W4_HSTORE(P)                                                                                           \
  {                                                                                                            \
    _Pragma("unroll") for (int k_ = 0; k_ < Q::NP; ++k_) {                                                     \
      w4_xwr2<(P)*Q::HBYTES>(hws[k_], hreg[k_][0], hreg[k_][1]);                                               \
      w4_xwr2<(P)*Q::HBYTES + Q::PAIR * 8>(hws[k_], hreg[k_][2], hreg[k_][3]);                                 \
    }                                                                                                          \
  }
    // filter of this wave: k-group h of co-tile ct = [ct][h][wave][3 x dwordx4 per lane], co-tiles ct0 and ct0 + 1.  Raw ISA: the waits are mine (tools/check_wino4_isa.py)
    const unsigned ubase = (unsigned)(ct0 * (C >> 2)) * W4W_KGB + (unsigned)wave * (3u * 64u * 16u);
#define W4_LOADB(DST, HS)                                                                                      \
  {                                                                                                            \
    const unsigned so_ = (HS);        /* byte offset of the k-group in co-tile ct0, W4_PAST = none */          \
    const unsigned so1_ = so_ + ctstride;                                                                      \
    DST.q[0] = w4_gld4<0>(ruv, uvo, so_); DST.q[1] = w4_gld4<1024>(ruv, uvo, so_); DST.s[0] = w4w_gld1<2048>(ruv, uvo, so_);     \
    DST.q[2] = w4_gld4<0>(ruv, uvo, so1_); DST.q[3] = w4_gld4<1024>(ruv, uvo, so1_); DST.s[1] = w4w_gld1<2048>(ruv, uvo, so1_);  \
  }
    W4WB b0, b1;
    f32x4 hreg[Q::NP];
    W4_HLOADS(0u)
    W4_LOADB(b0, ubase)
    W4_CLK()      /* item top: halo + filter loads issued */
    w4_vm_landedH(hreg);
    W4_HSTORE(0)
    W4_HLOADS((unsigned)Q::SBYTES)                      // stage 1's pieces fly during the first transform
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_CLK()      /* own pieces of stage 0 in LDS */
    __builtin_amdgcn_s_barrier();
    W4_CLK()      /* everyone's */
    asm volatile("" ::: "memory");
    if (tpart == 0) w4_transform<0, 0, 1>(hb0, vw0);
    else if (tpart == 1) w4_transform<0, 1, 1>(hb0, vw0);
    else w4_transform<0, 2, 1>(hb0, vw0);
    asm volatile("" ::: "memory");
    w4_vm_landedH(hreg);
    W4_HSTORE(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_CLK()      /* stage 0 transformed */
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    W4_CLK()      /* K loop starts */

    f32x4 acc[3][W4W_NT];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int nt = 0; nt < W4W_NT; ++nt) acc[pl][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 18 MFMAs of a k-group in three groups of 6 (one frequency point each); H0 / H1 / H2: the vector-memory
    // instruction issued behind each group (conv_wino4.hip: spread, not a burst behind the barrier)
#define W4_MUL6(PL, B)                                                                                         \
  _Pragma("unroll") for (int nt = 0; nt < W4W_NT; ++nt)                                                        \
      acc[PL][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[PL], w4w_bval(B, nt / 3, (PL)*3 + nt % 3),             \
                                                         acc[PL][nt], 0, 0, 0);
#define W4_MUL(P, G, B, H0, H1, H2)                                                                            \
  {                                                                                                            \
    float av_[3];                                                                                              \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) av_[pl] = va0[((P) ? W4_VBYTES / 4 : 0) + (pl * 4 + (G)) * 64];  \
    W4_MUL6(0, B) __builtin_amdgcn_sched_barrier(0); H0 __builtin_amdgcn_sched_barrier(0);                     \
    W4_MUL6(1, B) __builtin_amdgcn_sched_barrier(0); H1 __builtin_amdgcn_sched_barrier(0);                     \
    W4_MUL6(2, B) __builtin_amdgcn_sched_barrier(0); H2 __builtin_amdgcn_sched_barrier(0);                     \
  }
    // the transform of stage s + 1, one third of the waves at each of three points of the stage -- each point where
    // only ONE filter buffer is live (the other's multiplies are issued, its next load is not): the register budget
#define W4_TRANS(P, PART)                                                                                      \
  if (s_ + 1 < S && tpart == (PART)) {                                                                         \
    __builtin_amdgcn_s_setprio(3);                                                                             \
    w4_transform<1 - (P), PART, 1>(hb0, vw0 + (unsigned)((1 - (P)) * W4_VBYTES));                              \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  }
    // Stage s (parity P): k-groups 4s .. 4s+3.  Wait group h is issued one multiply (18 MFMAs) before it is awaited,
    // alone in the queue -- except the last wait of the stage, which also retires the halo pieces of stage s + 2
    // issued behind the first MFMA groups of k-group 3: no wait needs a count.
#define W4_STAGE(P, SI)                                                                                        \
  {                                                                                                            \
    const int s_ = (SI);                                                                                       \
    const unsigned dst_ = s_ + 2 < S ? (unsigned)(s_ + 2) * (unsigned)Q::SBYTES : W4_PAST;                     \
    const unsigned u0_ = ubase + (unsigned)(4 * s_) * W4W_KGB;                                                 \
    const unsigned bn_ = s_ + 1 < S ? u0_ + 4u * W4W_KGB : W4_PAST;                                            \
    w4w_vm_landedB(b0);                                                                                        \
    W4_CLK() /* 0: k-group 4s landed */                                                                        \
    W4_TRANS(P, 0)                                                                                             \
    W4_LOADB(b1, u0_ + W4W_KGB)                                                                                \
    W4_CLK() /* 1: (transform third 0 +) loads issued */                                                       \
    W4_MUL(P, 0, b0, , , )                                                                                     \
    W4_CLK() /* 2: k-group 0 multiplies issued */                                                              \
    w4w_vm_landedB(b1);                                                                                        \
    W4_CLK() /* 3: k-group 4s+1 landed */                                                                      \
    W4_TRANS(P, 1)                                                                                             \
    W4_LOADB(b0, u0_ + 2u * W4W_KGB)                                                                           \
    W4_MUL(P, 1, b1, , , )                                                                                     \
    W4_CLK() /* 4: (transform third 1 +) k-group 1 multiplies issued */                                        \
    w4w_vm_landedB(b0);                                                                                        \
    W4_LOADB(b1, u0_ + 3u * W4W_KGB)                                                                           \
    W4_MUL(P, 2, b0, , , )                                                                                     \
    W4_CLK() /* 5: k-group 2 multiplies issued */                                                              \
    w4w_vm_landedB(b1);                                                                                        \
    W4_TRANS(P, 2)                                                                                             \
    W4_LOADB(b0, bn_)                                                                                          \
    W4_MUL(P, 3, b1, W4_HLOAD(0, dst_), W4_HLOAD(1, dst_), )                                                   \
    W4_CLK() /* 6: (transform third 2 +) k-group 3 multiplies issued */                                        \
    w4_vm_landedH(hreg);             /* the pieces of stage s + 2 and k-group 4s+4 */                          \
    W4_HSTORE(P)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    W4_CLK() /* 7: own pieces of stage s + 2 in LDS, V writes done */                                          \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("" ::: "memory");                                                                             \
    W4_CLK() /* 8: past the barrier */                                                                         \
  }
    for (int s = 0; s + 1 < S; s += 2) {
      W4_STAGE(0, s)
      W4_STAGE(1, s + 1)
    }
    if (S & 1) W4_STAGE(0, S - 1)            // (every item starts at parity 0)
    // the loads past the end: tied to the wait (conv_wino4.hip)
    w4w_vm_landedB(b0);
    w4w_vm_landedB(b1);
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

    // ---- item end: per co-tile, accumulators -> LDS -> one (tile, co) per lane -> Y = A^T M A -> epilogue
    // exchange + output (conv_wino4.hip): this lane finishes tile 4 (wave & 3) + (lane >> 4), co 16 (wave >> 2) + li.
    // Addresses from an opaque copy of the lane id: computed here, not held in registers over the K loop.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int li_e = lane_e & 15, kq_e = lane_e >> 4;
    const int ont = wave >> 2, okq = wave & 3;
    const unsigned xhi = (unsigned)(li_e >> 3);
    const unsigned xw0 = lds0 + (unsigned)((3 * wave) * 3 * 1024 + lane_e * 16);
    const unsigned xwa = xw0 + 8u * xhi, xwb = xw0 + 8u - 8u * xhi;
    const unsigned xr0 = lds0 + (unsigned)((ont * 64 + okq * 16 + li_e) * 16) + (((unsigned)kq_e + 2u * xhi) & 3u) * 4u;
    const int tile = 4 * okq + kq_e;
    const int ty = tile >> 2, tx = tile & 3;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int cch = (ct0 + r) * W4_CO + ont * 16 + li_e;
      const float sc = a.scale[cch];
      const float sh = a.shift[cch];
      const unsigned vo = (unsigned)((((n * a.Ho + y0 + 4 * ty) * a.Wo + x0 + 4 * tx) * Co + cch) * 4);
      float rv[4][4];
#pragma unroll
      for (int oa = 0; oa < 4; ++oa)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
          rv[oa][ob] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     rr, has_res ? vo : EGN_OOB, oa * rowpitch + ob * colpitch, 0));
#define W4_XW(SL, V) w4_xwr2<(SL)*1024>(xwa, (V)[0], (V)[1]); w4_xwr2<(SL)*1024>(xwb, (V)[2], (V)[3]);
      W4_XW(0, acc[0][3 * r + 0]) W4_XW(1, acc[0][3 * r + 1]) W4_XW(2, acc[0][3 * r + 2])
      W4_XW(3, acc[1][3 * r + 0]) W4_XW(4, acc[1][3 * r + 1]) W4_XW(5, acc[1][3 * r + 2])
      W4_XW(6, acc[2][3 * r + 0]) W4_XW(7, acc[2][3 * r + 1]) W4_XW(8, acc[2][3 * r + 2])
#undef W4_XW
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      W4_CLK()    /* round: accumulators written */
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      W4_CLK()    /* round: exchange barrier passed */
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
#pragma unroll
      for (int oa = 0; oa < 4; ++oa) {
        float yo[4];
        w4_at(yc[oa], yo);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
          const float v = fmaxf(__builtin_fmaf(yo[ob], sc, sh) + rv[oa][ob], act_lo);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, vo, oa * rowpitch + ob * colpitch, 0);
        }
      }
      asm volatile("" ::: "memory");
      W4_CLK()    /* round: output transform done, stores issued */
      __builtin_amdgcn_s_barrier();      // the exchange buffer is free again (next round / next item's halo)
      asm volatile("" ::: "memory");
      W4_CLK()    /* round: end */
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
__global__ __launch_bounds__(W4_NTH, 1) void conv_wino4w_kernel(ConvArgs a) { w4w_body<ABL>(a); }

bool egn_conv_wino4w_applies(const ConvArgs& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 16 == 0 && a.Cin >= 32 && a.cs_in == a.Cin &&
         a.Cout % (2 * W4_CO) == 0 && a.cs_out == a.Cout && !a.out_nchw && a.Ho % 16 == 0 && a.Wo % 16 == 0 &&
         !(a.act & EGN_ACT_RES_AFTER) &&
         ((a.act & EGN_ACT_MASK) == EGN_ACT_NONE || (a.act & EGN_ACT_MASK) == EGN_ACT_RELU);
}

static unsigned w4w_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

template <int ABL>
static int wino4w_launch(ConvArgs a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];
  void (*kern)(ConvArgs) = &conv_wino4w_kernel<ABL>;
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 512));
  }
  const int ncp = a.Cout / (2 * W4_CO);
  const int nreg = a.tiles_x * a.tiles_y * a.N;
  const int imode = w4_item_mode(ncp);
  const int nwork = w4_item_count(imode, nreg, ncp, 1);
  if ((unsigned long long)nwork * (unsigned)(8 * ncp) >= 0x100000000ull ||
      (unsigned long long)(nreg + 8) * (unsigned)(a.tiles_x * a.tiles_y) >= 0x100000000ull)
    return EGN_E_BADARG;
  a.mg_nct = imode == 1 ? 0u : w4w_magic(imode == 2 ? ncp / 8 : ncp);
  a.mg_txy = w4w_magic(a.tiles_x * a.tiles_y);
  a.mg_tx = w4w_magic(a.tiles_x);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  int cap = cus / (8 * ncp) * (8 * ncp);
  if (cap <= 0) cap = 8 * ncp;
  const int grid = nwork < cap ? nwork : cap;               // one block per CU, whole XCD rounds
  hipLaunchKernelGGL(kern, dim3(grid), dim3(W4_NTH), lds, stream, a);
  return (int)hipGetLastError();
}
int egn_conv_launch_wino4w(ConvArgs a, size_t lds, int abl, hipStream_t stream) {
  if (!egn_conv_wino4w_applies(a)) return EGN_E_BADARG;
  switch (abl) {
    case 0: return wino4w_launch<0>(a, lds, stream);
#ifdef EGN_PROBES
    case 64: return wino4w_launch<64>(a, lds, stream);
#endif
    default: return EGN_E_BADARG;
  }
}
