// conv_wino4h.hip -- fused Winograd F(4x4,3x3) in HALF-size blocks, two independent blocks per CU [round 6].
//
// OUTCOME: measured NEGATIVE in both forms, kept for probe builds only (-DEGN_PROBES):
//   conv_wino4h_kernel (two 6-wave workgroups per CU)   64.3 us against conv_wino4_kernel's 51.1 on 48 -> 48 @ 64 x 64, 64 crops:
//                      the two workgroups of a CU are never resident together (profiles/r6_wino4h_timeline.txt);
//   conv_wino4d_kernel (two halves of one 12-wave workgroup, LDS-counter barriers)   52.7 us; 118 000 cycles per CU against
//                      103 000 (profiles/r6_wino4d_timeline.txt).  The start skew (EGN_W4H_SKEW, default 0) changes nothing.
// What follows is the design as it was built.
//
// Why.  An item of conv_wino4_kernel on the 48-channel branch (48 -> 48 @ 64 x 64: 24 % of the forward's kernel time)
// spends 11 % of its cycles in the prologue (first halo pieces, first transform) and 21 % in the item end (exchange,
// Y = A^T M A, stores) with the matrix pipe idle (profiles/r3_wino4_timeline_v2.txt; ablation: -28 % without the item
// end, profiles/r4_wino4_experiments.txt item 6) -- one 12-wave block per CU runs them in lock step.  conv_pw_kernel and
// conv_s2r_kernel (round 5) went from 52-58 % to 65-72 % MFMA busy with "several small independent blocks per CU": one
// block's item end runs under another's MFMAs.  This is that recipe for F(4x4,3x3):
//   * block = 6 waves, item = ONE m-tile (16 tiles = a 16 x 16 pixel region) x 48 output channels, 8-channel stages;
//     wave w owns frequency points 6w .. 6w+5 for the three co sub-tiles: 18 MFMAs per 4-channel k-group, 72 accumulator
//     registers -- per wave the same work per stage as conv_wino4_kernel (36 MFMAs, one third of a k-group's transform,
//     two halo pieces), per block half of it;
//   * LDS 62 KB (two V buffers [36 points][2 k-groups][64] = 18 KB, two halo buffers of 13 KB in GEO 0's store-conflict-
//     free order on GEO 1's region): two blocks per CU, 168 VGPRs = three waves per SIMD as before;
//   * the filter is read in wino4_pack.h's layout as it lies: the wave's 6 points are the slices of the 12-wave
//     kernel's waves 2w and 2w+1 (values 0..7 as two dwordx4, value 8 as a dword: 18 registers per buffer);
//   * item end: the 108 KB exchange image of an m-tile does not fit twice into a CU -- two rounds of 54 KB
//     ([point][6 units of 16 lanes][float4]: a unit = (co sub-tile, tile quad) = a quarter C fragment; round r holds
//     units 6r .. 6r+5, i.e. 1.5 co sub-tiles), reader wave rw finishes unit 6r + rw: every lane one (tile, co).
// Phase: two blocks that start together stay in phase (profiles/r3_wino9_kq2_timeline.txt) and would idle the pipe
// together.  The second block of a CU (its LDS allocation does not start at 0: HW_REG_LDS_ALLOC) therefore starts
// `skew` x 64 cycles late (ConvArgs::spix_off, set by the launcher from EGN_W4H_SKEW; no value helped).
// Every vector-memory wait is vmcnt(0) (conv_wino4.hip).
// Reference: the 3x3 stride-1 convolutions of libs/model/heatmapModel/hrnet.py (BasicBlock :49-76).
#include <stdlib.h>

#include "conv_wino4.h"

namespace {
typedef W4G<3> QH;
constexpr int W4H_NW = 6, W4H_NTH = 64 * W4H_NW;
constexpr int W4H_VBYTES = 36 * QH::VPT;                       // 18 KB
constexpr int W4H_V0 = 0, W4H_H0 = 2 * W4H_VBYTES;
constexpr int W4H_LDS = W4H_H0 + 2 * QH::HBYTES;               // 62 KB
constexpr int W4H_XPT = 6 * 256;                               // exchange bytes per point and round
static_assert(36 * W4H_XPT <= W4H_LDS, "the exchange reuses the stage buffers");
static_assert(QH::QPP * QH::RH * QH::RWP <= QH::NP * W4H_NW * 64, "the load pieces of the waves cover the halo");
static_assert(QH::RH * QH::XD + 2 * QH::XD <= QH::PLANE && 3 * QH::PAIR + 3 * QH::PLANE + 20 * QH::XD <= QH::HSLOT &&
                  QH::HSLOT * 8 <= QH::HBYTES,
              "every load lane stores to the natural slot of its element (no parking area)");
constexpr unsigned W4H_KGB = W4_UKG * 4u;                      // filter bytes of one (co-tile, k-group)

struct W4HB {
  f32x4 q[4];       // [2 o + (p >> 2)][p & 3], p < 8: slice o of the 12-wave layout (points 6w + 3o .. + 2)
  float s[2];       // value 8 of slice o
};
}  // namespace

__device__ __forceinline__ void w4h_vm_landedB(W4HB& b) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(b.q[0]), "+v"(b.q[1]), "+v"(b.q[2]), "+v"(b.q[3]), "+v"(b.s[0]), "+v"(b.s[1]));
}
template <int OFF>
__device__ __forceinline__ float w4h_gld1(u32x4 rsrc, unsigned voff, unsigned soff) {
  float v;
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF));
  return v;
}
// value of point pl (0..5) and co sub-tile nt: slice o = pl / 3, p = 3 (pl % 3) + nt
__device__ __forceinline__ float w4h_bval(const W4HB& b, int pl, int nt) {
  const int o = pl / 3, p = 3 * (pl % 3) + nt;
  return p < 8 ? b.q[2 * o + (p >> 2)][p & 3] : b.s[o];
}
template <int PT>
__device__ __forceinline__ float w4h_xrd(unsigned xr0, unsigned xr1) {
  if constexpr (PT < 18) return w4_lds<PT * W4H_XPT>(xr0);
  else return w4_lds<(PT - 18) * W4H_XPT>(xr1);
}

// Barrier of the six waves of a half (DUAL): arrive = one ds_add on the half's counter, wait = poll it.  The LDS serves a
// wave's operations in order, so the wave's earlier LDS writes (and reads) are done when its add is; a wave that has seen
// the count issues its reads after the writers' adds.  ~150-250 cycles against s_barrier's tens -- the price of two
// independent instruction streams in one workgroup.
__device__ __forceinline__ void w4h_subbar(unsigned addr, unsigned& target, int lane) {
  target += (unsigned)W4H_NW;
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
  for (;;) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - target) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

// ABL (probe builds): bit 6 s_memtime stamps of every wave (tools/wino4_clk.py), dumped into `res`.
// DUAL (conv_wino4d_kernel): ONE 12-wave workgroup hosts the two independent 6-wave blocks of a CU.  Two separate 6-wave
// workgroups of 168 VGPRs are never resident together: the dispatcher puts the waves of every workgroup on the SIMDs in the
// same cyclic order from the same start (2 / 2 / 1 / 1), and SIMDs 0 and 1 have room for three waves, not four (measured:
// profiles/r6_wino4h_timeline.txt -- 256 of 256 CUs ran their two blocks one after the other).  A 12-wave workgroup is
// placed 3 / 3 / 3 / 3; its halves (waves 0-5 / 6-11) have their own LDS image, their own items and their own barrier:
// s_barrier counts all 12 waves, so a half synchronises through a counter in LDS (w4h_subbar).
template <int ABL, bool DUAL>
__device__ __forceinline__ void w4h_body(const ConvArgs& a) {
  typedef QH Q;
  extern __shared__ float4 w4_smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = DUAL ? (wave_all >= W4H_NW ? 1 : 0) : 0;
  const int wave = wave_all - W4H_NW * half;
  const unsigned lds_wg = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_w4_t)w4_smem;
  const unsigned lds0 = lds_wg + (unsigned)(half * W4H_LDS);
  const float* smf = reinterpret_cast<const float*>(w4_smem) + half * (W4H_LDS / 4);
  // the halves' barrier counters (monotonic; zeroed below) behind the two LDS images
  const unsigned cnt_addr = lds_wg + (unsigned)(2 * W4H_LDS + 16 * half);
  unsigned bar_target = 0;
#define W4H_BAR()                                                  \
  {                                                                \
    if constexpr (DUAL) w4h_subbar(cnt_addr, bar_target, lane);    \
    else __builtin_amdgcn_s_barrier();                             \
  }
  const int li = lane & 15, kq = lane >> 4;
  const int tpart = wave >> 1;             // its third of the frequency rows of the input transform
  const int tg = wave & 1;                 // its k-group of the stage in the transform

  const int C = a.Cin, Co = a.Cout;
  const int nct = Co / W4_CO;
  const int S = C / 8;                     // stages of 8 channels

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

  // ---- halo loads: pieces wave, wave + 6; element e -> (pixel e / 2, channel quad e % 2).  Every lane stores to the
  // NATURAL slot of its element: the padded columns 18 / 19 fall on slots (x & 3) in {2, 3}, x >> 2 = 4 and the rows 18 / 19
  // on slots 90 .. 99 of a plane (PLANE = 100) -- never read by the transform; OOB offsets load zeros.
  unsigned hws[Q::NP];
#pragma unroll
  for (int k = 0; k < Q::NP; ++k) {
    const int e = (wave + W4H_NW * k) * 64 + lane;
    const int px = e / Q::QPP, hq = e % Q::QPP;
    const int hy = px / Q::RWP, hx = px - hy * Q::RWP;
    const int slot = 2 * hq * Q::PAIR + (hx & 3) * Q::PLANE + hy * Q::XD + (hx >> 2);
    hws[k] = lds0 + (unsigned)(W4H_H0 + slot * 8);
  }
  // ---- transform share: lane (tile li, channel 4 tg + kq of the stage)
  const unsigned hb0 = lds0 + (unsigned)(W4H_H0 + ((2 * tg + (kq >> 1)) * Q::PAIR + Q::tileslot(li, 0)) * 8 + (kq & 1) * 4);
  const unsigned vw0 = lds0 + (unsigned)(W4H_V0 + tg * 256 + lane * 4);         // V[pt][g][lane]
  // ---- multiply: A operands V[6 wave + pl][g][lane]
  const float* va0 = smf + (W4H_V0 / 4) + (6 * wave) * (Q::VPT / 4) + lane;
  const unsigned uvo = (unsigned)lane * 16u;

  const int regs_x = a.tiles_x, regs_xy = a.tiles_x * a.tiles_y;
  const int nreg = regs_xy * a.N;
  const int imode = w4_item_mode(nct);
  const int nwork = w4_item_count(imode, nreg, nct, 1);
  const int gsz = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  const float act_lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();
  const bool has_res = (ABL & 64) ? false : a.res != nullptr;
  const unsigned rowpitch = (unsigned)(a.Wo * Co) * 4u, colpitch = (unsigned)Co * 4u;

  constexpr int W4_NTK = 128;
  unsigned long long* sT = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(w4_smem) + (DUAL ? 2 * W4H_LDS + 64 : W4H_LDS));
  int ntk = 0;
#define W4_CLK()                                                                        \
  {                                                                                     \
    if constexpr ((ABL & 64) != 0) {                                                    \
      if (lane == 0 && ntk < W4_NTK) sT[wave_all * W4_NTK + ntk] = __builtin_readcyclecounter(); \
      ++ntk;                                                                            \
    }                                                                                   \
  }
  W4_CLK()
  // ---- phase skew: the block whose LDS allocation does not start at 0 shares its CU with an older block
  const unsigned lds_base = __builtin_amdgcn_s_getreg((12 - 1) << 11 | 0 << 6 | 6) & 0xfffu;      // HW_REG_LDS_ALLOC: LDS_BASE
  if constexpr (DUAL) {
    if (tid < 8) reinterpret_cast<unsigned*>(reinterpret_cast<char*>(w4_smem) + 2 * W4H_LDS)[tid] = 0u;
    __syncthreads();             // (the only workgroup barrier of the kernel body)
  }
  if (a.spix_off > 0 && (DUAL ? half != 0 : lds_base != 0u)) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long d = (unsigned long long)a.spix_off * 64ull;
    while (__builtin_readcyclecounter() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
  W4_CLK()
  // DUAL: the second halves are blocks gsz .. 2 gsz - 1 of a virtual grid (gsz is a multiple of 8: the same XCD)
  for (int w = blockIdx.x + half * gsz; w < nwork; w += (DUAL ? 2 : 1) * gsz) {
    const unsigned wi = (unsigned)__builtin_amdgcn_readfirstlane(w);
    const unsigned xq = wi & 7u, q_ = wi >> 3;
    const unsigned qq = w4_udiv(q_, a.mg_nct);
    int reg, ct;
    if (imode == 0) {
      ct = (int)(q_ - qq * (unsigned)nct);
      reg = (int)(qq * 8u + xq);
    } else if (imode == 1) {
      const unsigned lg = (unsigned)nct >> 1;
      reg = (int)(q_ * (8u >> lg) + (xq >> lg)); ct = (int)(xq & ((unsigned)nct - 1u));
    } else {
      reg = (int)qq; ct = (int)((q_ - qq * ((unsigned)nct >> 3)) * 8u + xq);
    }
    if (reg >= nreg) continue;
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
        const int e = (wave + W4H_NW * k) * 64 + lane_t;
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
    // filter of this wave: k-group h = [ct][h][12-wave slice 2w, 2w + 1][3 x dwordx4 per lane] -- 6 KB in a row.
    // Raw ISA: the waits are mine (tools/check_wino4_isa.py)
    const unsigned ubase = (unsigned)(ct * (C >> 2)) * W4H_KGB + (unsigned)wave * (6u * 64u * 16u);
#define W4_LOADB(DST, HS)                                                                                      \
  {                                                                                                            \
    const unsigned so_ = (HS);        /* byte offset of the k-group, W4_PAST = none */                         \
    const unsigned so1_ = so_ + 3072u; /* the second slice (the immediate offset of a buffer instruction has 12 bits) */ \
    DST.q[0] = w4_gld4<0>(ruv, uvo, so_); DST.q[1] = w4_gld4<1024>(ruv, uvo, so_); DST.s[0] = w4h_gld1<2048>(ruv, uvo, so_);     \
    DST.q[2] = w4_gld4<0>(ruv, uvo, so1_); DST.q[3] = w4_gld4<1024>(ruv, uvo, so1_); DST.s[1] = w4h_gld1<2048>(ruv, uvo, so1_);  \
  }
    W4HB b0, b1;
    f32x4 hreg[Q::NP];
    W4_HLOADS(0u)
    W4_LOADB(b0, ubase)
    W4_CLK()      /* item top: halo + filter loads issued */
    w4_vm_landedH(hreg);
    W4_HSTORE(0)
    W4_HLOADS((unsigned)Q::SBYTES)                      // stage 1's pieces fly during the first transform
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_CLK()      /* own pieces of stage 0 in LDS */
    W4H_BAR()
    W4_CLK()      /* everyone's */
    asm volatile("" ::: "memory");
    if (tpart == 0) w4_transform<0, 0, 3>(hb0, vw0);
    else if (tpart == 1) w4_transform<0, 1, 3>(hb0, vw0);
    else w4_transform<0, 2, 3>(hb0, vw0);
    asm volatile("" ::: "memory");
    w4_vm_landedH(hreg);
    W4_HSTORE(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_CLK()      /* stage 0 transformed */
    W4H_BAR()
    asm volatile("" ::: "memory");
    W4_CLK()      /* K loop starts */

    f32x4 acc[6][3];
#pragma unroll
    for (int pl = 0; pl < 6; ++pl)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[pl][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 18 MFMAs of a k-group in three groups of 6 (two frequency points each); H0 / H1 / H2: the vector-memory
    // instruction issued behind each group (conv_wino4.hip: spread, not a burst behind the barrier)
#define W4_MUL6(J, B)                                                                                          \
  _Pragma("unroll") for (int x_ = 0; x_ < 2; ++x_) _Pragma("unroll") for (int nt = 0; nt < 3; ++nt)            \
      acc[2 * (J) + x_][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[2 * (J) + x_], w4h_bval(B, 2 * (J) + x_, nt), \
                                                                   acc[2 * (J) + x_][nt], 0, 0, 0);
#define W4_MUL(P, G, B, H0, H1, H2)                                                                            \
  {                                                                                                            \
    float av_[6];                                                                                              \
    _Pragma("unroll") for (int pl = 0; pl < 6; ++pl)                                                           \
        av_[pl] = va0[((P) ? W4H_VBYTES / 4 : 0) + pl * (Q::VPT / 4) + (G)*64];                                \
    W4_MUL6(0, B) __builtin_amdgcn_sched_barrier(0); H0 __builtin_amdgcn_sched_barrier(0);                     \
    W4_MUL6(1, B) __builtin_amdgcn_sched_barrier(0); H1 __builtin_amdgcn_sched_barrier(0);                     \
    W4_MUL6(2, B) __builtin_amdgcn_sched_barrier(0); H2 __builtin_amdgcn_sched_barrier(0);                     \
  }
#define W4_TRANS(P, PART)                                                                                      \
  if (s_ + 1 < S && tpart == (PART)) {                                                                         \
    __builtin_amdgcn_s_setprio(3);                                                                             \
    w4_transform<1 - (P), PART, 3>(hb0, vw0 + (unsigned)((1 - (P)) * W4H_VBYTES));                             \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  }
    // Stage s (parity P): k-groups 2s, 2s+1 (conv_wino4.hip's GEO 0 stage with one m-tile and six points per wave)
#define W4_STAGE(P, SI)                                                                                        \
  {                                                                                                            \
    const int s_ = (SI);                                                                                       \
    const unsigned dst_ = s_ + 2 < S ? (unsigned)(s_ + 2) * (unsigned)Q::SBYTES : W4_PAST;                     \
    const unsigned u0_ = ubase + (unsigned)(2 * s_) * W4H_KGB;                                                 \
    const unsigned bn_ = s_ + 1 < S ? u0_ + 2u * W4H_KGB : W4_PAST;                                            \
    w4h_vm_landedB(b0);                                                                                        \
    W4_CLK() /* 0: k-group 2s landed */                                                                        \
    W4_TRANS(P, 0)                                                                                             \
    W4_LOADB(b1, u0_ + W4H_KGB)                                                                                \
    W4_CLK() /* 1: (transform third 0 +) loads issued */                                                       \
    W4_MUL(P, 0, b0, , , )                                                                                     \
    W4_CLK() /* 2: k-group 0 multiplies issued */                                                              \
    w4h_vm_landedB(b1);                                                                                        \
    W4_CLK() /* 3: k-group 2s+1 landed */                                                                      \
    W4_TRANS(P, 1)                                                                                             \
    W4_LOADB(b0, bn_)                                                                                          \
    W4_MUL(P, 1, b1, W4_HLOAD(0, dst_), W4_HLOAD(1, dst_), )                                                   \
    W4_TRANS(P, 2)                                                                                             \
    W4_CLK() /* 4: (transform third 1 +) k-group 1 multiplies issued (+ transform third 2) */                  \
    w4_vm_landedH(hreg);             /* the pieces of stage s + 2 and k-group 2s+2 */                          \
    W4_HSTORE(P)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    W4_CLK() /* 5: own pieces of stage s + 2 in LDS, V writes done */                                          \
    W4H_BAR()                                                                              \
    asm volatile("" ::: "memory");                                                                             \
    W4_CLK() /* 6: past the barrier */                                                                         \
  }
    for (int s = 0; s + 1 < S; s += 2) {     // (S is even: Cin % 16 == 0)
      W4_STAGE(0, s)
      W4_STAGE(1, s + 1)
    }
    // the loads past the end: tied to the wait (conv_wino4.hip)
    w4h_vm_landedB(b0);
    w4h_vm_landedB(b1);
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

    // ---- item end.  Unit U = 4 nt + kq (a quarter C fragment: tiles 4 kq .. 4 kq + 3 x 16 channels of sub-tile nt);
    // round r holds units 6 r .. 6 r + 5 at [point][U - 6 r][writer lane li] float4; reader wave rw takes unit 6 r + rw.
    // Addresses from an opaque copy of the lane id: computed here, not held in registers over the K loop.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int li_e = lane_e & 15, kq_e = lane_e >> 4;
    const unsigned xhi = (unsigned)(li_e >> 3);
    // writer: slot of (nt, kq) in its round; li >= 8 stores its float4 rotated by two dwords (conv_wino4.hip: the reader
    // takes one dword of 16 writer slots -- li and li + 8 would share a bank)
    const unsigned xw_ = lds0 + (unsigned)((6 * wave) * W4H_XPT + li_e * 16);
    const unsigned xw_n0 = xw_ + (unsigned)(kq_e * 256);                              // nt 0: round 0, units 0..3
    const unsigned xw_n1 = xw_ + (unsigned)((kq_e < 2 ? 4 + kq_e : kq_e - 2) * 256);  // nt 1: round 0 units 4, 5 | round 1 units 0, 1
    const unsigned xw_n2 = xw_ + (unsigned)((2 + kq_e) * 256);                        // nt 2: round 1, units 2..5
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int U = 6 * r + wave;                       // this wave's unit as a reader
      const int nt_r = U >> 2, kq_w = U & 3;
      const int tile = 4 * kq_w + kq_e;
      const int ty = tile >> 2, tx = tile & 3;
      const int cch = ct * W4_CO + nt_r * 16 + li_e;
      const float sc = a.scale[cch];
      const float sh = a.shift[cch];
      const unsigned vo = (unsigned)((((n * a.Ho + y0 + 4 * ty) * a.Wo + x0 + 4 * tx) * Co + cch) * 4);
      const unsigned xr0 = lds0 + (unsigned)(wave * 256 + li_e * 16) + (((unsigned)kq_e + 2u * xhi) & 3u) * 4u;
      float rv[4][4];
#pragma unroll
      for (int oa = 0; oa < 4; ++oa)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
          rv[oa][ob] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     rr, has_res ? vo : EGN_OOB, oa * rowpitch + ob * colpitch, 0));
#define W4_XW(ADDR, PL, V) w4_xwr2<(PL)*W4H_XPT>((ADDR) + 8u * xhi, (V)[0], (V)[1]); w4_xwr2<(PL)*W4H_XPT>((ADDR) + 8u - 8u * xhi, (V)[2], (V)[3]);
#define W4_XW6(ADDR, NT) W4_XW(ADDR, 0, acc[0][NT]) W4_XW(ADDR, 1, acc[1][NT]) W4_XW(ADDR, 2, acc[2][NT])    \
                         W4_XW(ADDR, 3, acc[3][NT]) W4_XW(ADDR, 4, acc[4][NT]) W4_XW(ADDR, 5, acc[5][NT])
      if (r == 0) {
        W4_XW6(xw_n0, 0)
        if (kq_e < 2) { W4_XW6(xw_n1, 1) }
      } else {
        if (kq_e >= 2) { W4_XW6(xw_n1, 1) }
        W4_XW6(xw_n2, 2)
      }
#undef W4_XW6
#undef W4_XW
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      W4_CLK()    /* round: accumulators written */
      W4H_BAR()
      asm volatile("" ::: "memory");
      W4_CLK()    /* round: exchange barrier passed */
      float yc[4][6];
#define W4_M(I, J) w4h_xrd<(I)*6 + (J)>(xr0, xr1)
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
      const unsigned xr1 = xr0 + 18u * (unsigned)W4H_XPT;
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
      W4H_BAR()      // the exchange buffer is free again (next round / next item's halo)
      asm volatile("" ::: "memory");
      W4_CLK()    /* round: end */
    }
  }
  if constexpr ((ABL & 64) != 0) {
    __syncthreads();
    constexpr int NWA = DUAL ? 2 * W4H_NW : W4H_NW;
    unsigned long long* out = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res)) +
                              (size_t)blockIdx.x * (NWA * W4_NTK + 2);
    for (int e = tid; e < NWA * W4_NTK; e += NWA * 64) out[2 + e] = sT[e];
    if (tid == 0) { out[0] = (unsigned long long)ntk; out[1] = ((unsigned long long)lds_base << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xfu) << 16) |
                             (__builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4) & 0xffffu); }      /* LDS base | XCC_ID | HW_ID */
  }
#undef W4_CLK
#undef W4H_BAR
}

template <int ABL>
__global__ __launch_bounds__(W4H_NTH, 3) void conv_wino4h_kernel(ConvArgs a) { w4h_body<ABL, false>(a); }
template <int ABL>
__global__ __launch_bounds__(2 * W4H_NTH, 1) void conv_wino4d_kernel(ConvArgs a) { w4h_body<ABL, true>(a); }

bool egn_conv_wino4h_applies(const ConvArgs& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 16 == 0 && a.cs_in == a.Cin &&
         a.Cout % W4_CO == 0 && a.cs_out == a.Cout && !a.out_nchw && a.Ho % 16 == 0 && a.Wo % 16 == 0 &&
         !(a.act & EGN_ACT_RES_AFTER) &&
         ((a.act & EGN_ACT_MASK) == EGN_ACT_NONE || (a.act & EGN_ACT_MASK) == EGN_ACT_RELU);
}
// (+ the stamp area of the ABL & 64 build); dual: both halves' images + their barrier counters
size_t egn_conv_wino4h_lds_bytes(int dual) { return dual ? 2 * W4H_LDS + 64 + 12 * 128 * 8 : W4H_LDS + 6 * 128 * 8; }

#ifdef EGN_PROBES
static unsigned w4h_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
// start delay of a CU's second block in units of 64 cycles (EGN_W4H_SKEW overrides: probes)
static int w4h_default_skew() {
  static int skew = -1;
  if (skew < 0) {
    const char* e = getenv("EGN_W4H_SKEW");
    skew = e ? atoi(e) : 0;
    if (skew < 0) skew = 0;
  }
  return skew;
}
#endif

#ifdef EGN_PROBES
template <int ABL, bool DUAL>
static int wino4h_launch(ConvArgs a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];
  void (*kern)(ConvArgs) = DUAL ? &conv_wino4d_kernel<ABL> : &conv_wino4h_kernel<ABL>;
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      DUAL ? 160 * 1024 - 512 : 80 * 1024));
  }
  const int nct = a.Cout / W4_CO;
  const int nreg = a.tiles_x * a.tiles_y * a.N;
  const int imode = w4_item_mode(nct);
  const int nwork = w4_item_count(imode, nreg, nct, 1);
  if ((unsigned long long)nwork * (unsigned)(8 * nct) >= 0x100000000ull ||
      (unsigned long long)(nreg + 8) * (unsigned)(a.tiles_x * a.tiles_y) >= 0x100000000ull)
    return EGN_E_BADARG;
  a.mg_nct = imode == 1 ? 0u : w4h_magic(imode == 2 ? nct / 8 : nct);
  a.mg_txy = w4h_magic(a.tiles_x * a.tiles_y);
  a.mg_tx = w4h_magic(a.tiles_x);
  a.spix_off = w4h_default_skew();
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  // blocks per CU: two 6-wave workgroups, or one 12-wave workgroup with two halves; whole XCD rounds of co-tiles
  int cap = (DUAL ? 1 : 2) * cus / (8 * nct) * (8 * nct);
  if (cap <= 0) cap = 8 * nct;
  // (DUAL: a workgroup's halves take items w and w + grid: half as many workgroups as items fill both halves)
  const int want = DUAL ? (nwork / 2 + 8 * nct - 1) / (8 * nct) * (8 * nct) : nwork;
  const int grid = want < cap ? want : cap;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(DUAL ? 2 * W4H_NTH : W4H_NTH), lds, stream, a);
  return (int)hipGetLastError();
}
#endif
// Probe builds only (python -m egonet_amd.build --probes): both forms measured SLOWER than conv_wino4_kernel on the layer
// they were built for (profiles/r6_wino4h_timeline.txt, r6_wino4d_timeline.txt) -- the product library does not compile them.
int egn_conv_launch_wino4h(ConvArgs a, size_t lds, int abl, int dual, hipStream_t stream) {
#ifdef EGN_PROBES
  if (!egn_conv_wino4h_applies(a)) return EGN_E_BADARG;
  switch (abl) {
    case 0: return dual ? wino4h_launch<0, true>(a, lds, stream) : wino4h_launch<0, false>(a, lds, stream);
    case 64: return dual ? wino4h_launch<64, true>(a, lds, stream) : wino4h_launch<64, false>(a, lds, stream);
    default: return EGN_E_BADARG;
  }
#else
  (void)a; (void)lds; (void)abl; (void)dual; (void)stream;
  return EGN_E_BADARG;
#endif
}
