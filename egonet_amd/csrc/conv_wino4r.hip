// conv_wino4r.hip -- fused Winograd F(4x4,3x3), "row owner" waves: ONE exchange round per item [round 6].
//
// OUTCOME: correct, measured SLOWER than conv_wino4_kernel (54.1 against 50.4 us on 48 -> 48 @ 64 x 64 at 64 crops, 216.7
// against 192.7 on 256 -> 48: profiles/r6_wino4r_probe.txt) -- probe builds only (-DEGN_PROBES).  The
// claim below that the doubled filter stream is free did not hold at twelve load instructions per wave and stage.
// What follows is the design as it was built.
//
// conv_wino4_kernel's item end takes 21 % of an item on the 48-channel branch (profiles/r3_wino4_timeline_v2.txt): a wave
// holds three frequency points of a row of the 6 x 6 grid for BOTH m-tiles, nothing of Y = A^T M A can be formed in its
// registers, so all 36 points of every (tile, channel) cross the LDS -- 108 KB per m-tile, two rounds of write / barrier /
// 36 reads / barrier.  Here a wave owns a whole ROW of the grid for ONE m-tile (waves 0-5: m-tile 0, rows 0-5; waves 6-11:
// m-tile 1) -- 6 points x 3 co sub-tiles x 1 m-tile, the same 72 accumulators and 36 MFMAs per stage -- and applies the
// row pass of the output transform to its own accumulators: R[i][b] = sum_j M[i][j] A^T[b][j], 4 values where there
// were 6.  What crosses the LDS is R: 24 values per (tile, channel) instead of 36, 144 KB for BOTH m-tiles -- it fits the
// CU once the stage buffers are dead -- so the item end is ONE round: 24 ds_write_b64 per lane, one barrier, 2 x 24 reads,
// the column pass Y[a][b] = sum_i A^T[a][i] R[i][b] (the same 40 + 40 instructions that followed the exchange before;
// the row pass's 120 took the place of the first column pass), epilogue, stores, one barrier.  Half the barriers, two
// thirds of the LDS traffic, the same VALU work.
// The K loop is conv_wino4_kernel's (16 x 32 pixel regions, 8-channel stages, the same halo order, transform thirds and V
// layout) with conv_wino4h.hip's multiply: the wave's six points are the slices 2 i and 2 i + 1 of wino4_pack.h's layout
// (values 0..7 as two dwordx4, value 8 as a dword: 18 registers per filter buffer); both m-tile groups read the same
// filter (twice the L2 filter stream of conv_wino4_kernel: free beside MFMAs, profiles/r4_wino4_experiments.txt item 6).
// Numerics: the two 1-D passes of the output transform run in the other order (rows, then columns) -- the same sums,
// associated differently: NOT bit-identical to conv_wino4_kernel, same error class (tests: 5e-4 against the direct
// convolution, the W48 reference fixtures through the table).  Every vector-memory wait is vmcnt(0).
// Reference: the 3x3 stride-1 convolutions of libs/model/heatmapModel/hrnet.py (BasicBlock :49-76).
#include <stdlib.h>

#include "conv_wino4.h"

namespace {
typedef W4G<0> QR;
constexpr unsigned W4R_KGB = W4_UKG * 4u;                 // filter bytes of one (co-tile, k-group)
constexpr int W4R_XWAVE = 12 * 1024;                     // exchange bytes of a writer wave: [nt 3][b 4][lane 64] float4
constexpr int W4R_XBYTES = 12 * W4R_XWAVE;               // 147 456 B
constexpr int W4R_LDS = W4R_XBYTES > w4_lds_bytes<0>() ? W4R_XBYTES : w4_lds_bytes<0>();
static_assert(W4R_LDS + 12 * 96 * 8 <= 160 * 1024 - 512, "the one-round exchange (and the stamp area) fit the CU");
static_assert(5 * W4R_XWAVE + 3 * 1024 < 65536, "ds_read_b32 immediates of a pass");

struct W4RB {
  f32x4 q[4];       // [2 o + (p >> 2)][p & 3], p < 8: slice o of the 12-wave layout (points 6 row + 3 o .. + 2)
  float s[2];       // value 8 of slice o
};
}  // namespace

__device__ __forceinline__ void w4r_vm_landedB(W4RB& b) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(b.q[0]), "+v"(b.q[1]), "+v"(b.q[2]), "+v"(b.q[3]), "+v"(b.s[0]), "+v"(b.s[1]));
}
template <int OFF>
__device__ __forceinline__ float w4r_gld1(u32x4 rsrc, unsigned voff, unsigned soff) {
  static_assert(OFF >= 0 && OFF < 4096, "12-bit immediate");
  float v;
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF));
  return v;
}
// value of column j (0..5) of the wave's row and co sub-tile nt: slice o = j / 3, p = 3 (j % 3) + nt
__device__ __forceinline__ float w4r_bval(const W4RB& b, int j, int nt) {
  const int o = j / 3, p = 3 * (j % 3) + nt;
  return p < 8 ? b.q[2 * o + (p >> 2)][p & 3] : b.s[o];
}

// Row pass of the output transform for co sub-tile NT, in the wave's own accumulators, and its four exchange slots:
// R[b] = sum_j acc[j][NT][e] A^T[b][j] for the four tile elements e of the C fragment (10 instructions each).
template <int NT>
__device__ __forceinline__ void w4r_row_pass(const f32x4 (&acc)[6][3], unsigned xwa, unsigned xwb) {
  float rb[4][4];      // [b][e]
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float m[6], r4[4];
#pragma unroll
    for (int j = 0; j < 6; ++j) m[j] = acc[j][NT][e];
    w4_at(m, r4);
#pragma unroll
    for (int b = 0; b < 4; ++b) rb[b][e] = r4[b];
  }
  w4_xwr2<(NT * 4 + 0) * 1024>(xwa, rb[0][0], rb[0][1]); w4_xwr2<(NT * 4 + 0) * 1024>(xwb, rb[0][2], rb[0][3]);
  w4_xwr2<(NT * 4 + 1) * 1024>(xwa, rb[1][0], rb[1][1]); w4_xwr2<(NT * 4 + 1) * 1024>(xwb, rb[1][2], rb[1][3]);
  w4_xwr2<(NT * 4 + 2) * 1024>(xwa, rb[2][0], rb[2][1]); w4_xwr2<(NT * 4 + 2) * 1024>(xwb, rb[2][2], rb[2][3]);
  w4_xwr2<(NT * 4 + 3) * 1024>(xwa, rb[3][0], rb[3][1]); w4_xwr2<(NT * 4 + 3) * 1024>(xwb, rb[3][2], rb[3][3]);
}

// ABL (probe builds): bit 6 s_memtime stamps of every wave (tools/wino4_clk.py), dumped into `res`.
template <int ABL>
__device__ __forceinline__ void w4r_body(const ConvArgs& a) {
  typedef QR Q;
  extern __shared__ float4 w4_smem[];
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_w4_t)w4_smem;
  const float* smf = reinterpret_cast<const float*>(w4_smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int tpart = wave >> 2;             // its third of the frequency rows of the input transform
  const int tw = wave & 3;                 // its share of the transform: m-tile tw >> 1, k-group tw & 1
  const int mtw = wave >= 6 ? 1 : 0;       // multiply / row pass: its m-tile ...
  const int row = wave - 6 * mtw;          // ... and its row of the 6 x 6 grid (points 6 row .. 6 row + 5)

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

  // ---- halo loads (conv_wino4.hip, geometry 0): pieces wave, wave + 12; element e -> (pixel e / 2, channel quad e % 2);
  // the 96 elements past the halo (rows 18 / 19) park their zeros in the unused tail of the buffer
  unsigned hws[Q::NP], hws2[Q::NP];
#pragma unroll
  for (int k = 0; k < Q::NP; ++k) {
    const int e = (wave + W4_NW * k) * 64 + lane;
    const int px = e / Q::QPP, hq = e % Q::QPP;
    const int hy = px / Q::RWP, hx = px - hy * Q::RWP;
    const bool ok = hy < Q::RH && hx < Q::RW;
    const int slot = 2 * hq * Q::PAIR + (hx & 3) * Q::PLANE + hy * Q::XD + (hx >> 2);
    hws[k] = lds0 + (unsigned)(W4_H0 + (ok ? slot * 8 : Q::HSLOT * 8 + lane * 8));
    hws2[k] = lds0 + (unsigned)(W4_H0 + (ok ? (slot + Q::PAIR) * 8 : Q::HSLOT * 8 + 512 + lane * 8));
  }
  // ---- transform share: lane (tile li of m-tile tw >> 1, channel 4 (tw & 1) + kq of the stage)
  const unsigned hb0 = lds0 + (unsigned)(W4_H0 + ((2 * (tw & 1) + (kq >> 1)) * Q::PAIR + Q::tileslot(li, tw)) * 8 + (kq & 1) * 4);
  const unsigned vw0 = lds0 + (unsigned)(W4_V0 + tw * 256 + lane * 4);        // V[pt][mt * 2 + g][lane]
  // ---- multiply: A operands V[6 row + j][mtw][g][lane]
  const float* va0 = smf + (W4_V0 / 4) + (6 * row) * 256 + mtw * 128 + lane;
  const unsigned uvo = (unsigned)lane * 16u;

  const int regs_x = a.tiles_x, regs_xy = a.tiles_x * a.tiles_y;
  const int nreg = regs_xy * a.N;
  const int imode = w4_item_mode(nct);
  const int nwork = w4_item_count(imode, nreg, nct, 1);
  const int gsz = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  const float act_lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();
  const bool has_res = (ABL & 64) ? false : a.res != nullptr;
  const unsigned rowpitch = (unsigned)(a.Wo * Co) * 4u, colpitch = (unsigned)Co * 4u;

  constexpr int W4_NTK = 96;
  unsigned long long* sT = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(w4_smem) + W4R_LDS);
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
      w4_xwr2<(P)*Q::HBYTES>(hws2[k_], hreg[k_][2], hreg[k_][3]);                                              \
    }                                                                                                          \
  }
    // filter of this wave: k-group h = [ct][h][12-wave slices 2 row, 2 row + 1][3 x dwordx4 per lane] -- 6 KB in a row.
    // Raw ISA: the waits are mine (tools/check_wino4_isa.py)
    const unsigned ubase = (unsigned)(ct * (C >> 2)) * W4R_KGB + (unsigned)row * (6u * 64u * 16u);
#define W4_LOADB(DST, HS)                                                                                      \
  {                                                                                                            \
    const unsigned so_ = (HS);        /* byte offset of the k-group, W4_PAST = none */                         \
    const unsigned so1_ = so_ + 3072u; /* the second slice (12-bit immediates) */                              \
    DST.q[0] = w4_gld4<0>(ruv, uvo, so_); DST.q[1] = w4_gld4<1024>(ruv, uvo, so_); DST.s[0] = w4r_gld1<2048>(ruv, uvo, so_);     \
    DST.q[2] = w4_gld4<0>(ruv, uvo, so1_); DST.q[3] = w4_gld4<1024>(ruv, uvo, so1_); DST.s[1] = w4r_gld1<2048>(ruv, uvo, so1_);  \
  }
    W4RB b0, b1;
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
    if (tpart == 0) w4_transform<0, 0, 0>(hb0, vw0);
    else if (tpart == 1) w4_transform<0, 1, 0>(hb0, vw0);
    else w4_transform<0, 2, 0>(hb0, vw0);
    asm volatile("" ::: "memory");
    w4_vm_landedH(hreg);
    W4_HSTORE(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_CLK()      /* stage 0 transformed */
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    W4_CLK()      /* K loop starts */

    f32x4 acc[6][3];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 18 MFMAs of a k-group in three groups of 6 (two columns of the row each); H0 / H1 / H2: the vector-memory
    // instruction issued behind each group (conv_wino4.hip: spread, not a burst behind the barrier)
#define W4_MUL6(J, B)                                                                                          \
  _Pragma("unroll") for (int x_ = 0; x_ < 2; ++x_) _Pragma("unroll") for (int nt = 0; nt < 3; ++nt)            \
      acc[2 * (J) + x_][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[2 * (J) + x_], w4r_bval(B, 2 * (J) + x_, nt), \
                                                                   acc[2 * (J) + x_][nt], 0, 0, 0);
#define W4_MUL(P, G, B, H0, H1, H2)                                                                            \
  {                                                                                                            \
    float av_[6];                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 6; ++j) av_[j] = va0[((P) ? W4_VBYTES / 4 : 0) + j * 256 + (G)*64];   \
    W4_MUL6(0, B) __builtin_amdgcn_sched_barrier(0); H0 __builtin_amdgcn_sched_barrier(0);                     \
    W4_MUL6(1, B) __builtin_amdgcn_sched_barrier(0); H1 __builtin_amdgcn_sched_barrier(0);                     \
    W4_MUL6(2, B) __builtin_amdgcn_sched_barrier(0); H2 __builtin_amdgcn_sched_barrier(0);                     \
  }
#define W4_TRANS(P, PART)                                                                                      \
  if (s_ + 1 < S && tpart == (PART)) {                                                                         \
    __builtin_amdgcn_s_setprio(3);                                                                             \
    w4_transform<1 - (P), PART, 0>(hb0, vw0 + (unsigned)((1 - (P)) * W4_VBYTES));                              \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  }
    // Stage s (parity P): k-groups 2s, 2s+1 -- conv_wino4.hip's stage; each transform third where one filter buffer is live
#define W4_STAGE(P, SI)                                                                                        \
  {                                                                                                            \
    const int s_ = (SI);                                                                                       \
    const unsigned dst_ = s_ + 2 < S ? (unsigned)(s_ + 2) * (unsigned)Q::SBYTES : W4_PAST;                     \
    const unsigned u0_ = ubase + (unsigned)(2 * s_) * W4R_KGB;                                                 \
    const unsigned bn_ = s_ + 1 < S ? u0_ + 2u * W4R_KGB : W4_PAST;                                            \
    w4r_vm_landedB(b0);                                                                                        \
    W4_CLK() /* 0: k-group 2s landed */                                                                        \
    W4_TRANS(P, 0)                                                                                             \
    W4_LOADB(b1, u0_ + W4R_KGB)                                                                                \
    W4_CLK() /* 1: (transform third 0 +) loads issued */                                                       \
    W4_MUL(P, 0, b0, , , )                                                                                     \
    W4_CLK() /* 2: k-group 0 multiplies issued */                                                              \
    w4r_vm_landedB(b1);                                                                                        \
    W4_CLK() /* 3: k-group 2s+1 landed */                                                                      \
    W4_TRANS(P, 1)                                                                                             \
    W4_TRANS(P, 2)                                                                                             \
    W4_LOADB(b0, bn_)                                                                                          \
    W4_MUL(P, 1, b1, W4_HLOAD(0, dst_), W4_HLOAD(1, dst_), )                                                   \
    W4_CLK() /* 4: (transform thirds 1 / 2 +) k-group 1 multiplies issued */                                   \
    w4_vm_landedH(hreg);             /* the pieces of stage s + 2 and k-group 2s+2 */                          \
    W4_HSTORE(P)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    W4_CLK() /* 5: own pieces of stage s + 2 in LDS, V writes done */                                          \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("" ::: "memory");                                                                             \
    W4_CLK() /* 6: past the barrier */                                                                         \
  }
    for (int s = 0; s + 1 < S; s += 2) {     // (S is even: Cin % 16 == 0)
      W4_STAGE(0, s)
      W4_STAGE(1, s + 1)
    }
    // the loads past the end: tied to the wait (conv_wino4.hip)
    w4r_vm_landedB(b0);
    w4r_vm_landedB(b1);
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

    // ---- item end, ONE round.  Row pass in registers: for co sub-tile nt and tile element e of the C fragment,
    // R[b] = sum_j acc[j][nt][e] A^T[b][j].  Exchange [writer wave (m-tile, row i)][nt][b][writer lane] float4 (the four
    // tiles 4 kq .. 4 kq + 3 of channel li).  Addresses from an opaque copy of the lane id (not held over the K loop).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int li_e = lane_e & 15, kq_e = lane_e >> 4;
    const unsigned xhi = (unsigned)(li_e >> 3);
    // (writers with li >= 8 store their float4 rotated by two dwords: the reader takes one dword of 16 writer slots,
    // li and li + 8 would share a bank -- conv_wino4.hip)
    const unsigned xw0 = lds0 + (unsigned)(wave * W4R_XWAVE + lane_e * 16);
    const unsigned xwa = xw0 + 8u * xhi, xwb = xw0 + 8u - 8u * xhi;
    w4r_row_pass<0>(acc, xwa, xwb);
    w4r_row_pass<1>(acc, xwa, xwb);
    w4r_row_pass<2>(acc, xwa, xwb);
    // this lane finishes, per m-tile (pass), tile 4 (wave & 3) + (lane >> 4), channel 16 (wave >> 2) + li
    const int ont = wave >> 2, okq = wave & 3;
    const int tile = 4 * okq + kq_e;
    const int cch = ct * W4_CO + ont * 16 + li_e;
    const float sc = a.scale[cch];
    const float sh = a.shift[cch];
    unsigned vo[2];
    float rv[2][4][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int ty = 2 * mt + (tile >> 3), tx = tile & 7;
      vo[mt] = (unsigned)((((n * a.Ho + y0 + 4 * ty) * a.Wo + x0 + 4 * tx) * Co + cch) * 4);
#pragma unroll
      for (int oa = 0; oa < 4; ++oa)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
          rv[mt][oa][ob] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                         rr, has_res ? vo[mt] : EGN_OOB, oa * rowpitch + ob * colpitch, 0));
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the exchange writes (the residual loads stay in flight)
    W4_CLK()    /* row pass done, R written */
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    W4_CLK()    /* exchange barrier passed */
    // reader: R[i][b] of (m-tile mt, row i) = writer wave 6 mt + i, unit (nt = ont, b), writer lane 16 okq + li, element kq
    const unsigned xr0 = lds0 + (unsigned)(ont * 4096 + (okq * 16 + li_e) * 16) + (((unsigned)kq_e + 2u * xhi) & 3u) * 4u;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const unsigned xr = xr0 + (unsigned)(mt * 6 * W4R_XWAVE);
      float ycol[4][4];      // [a][b]
#define W4_R(I, B) w4_lds<(I)*W4R_XWAVE + (B)*1024>(xr)
#define W4_COL2(B0)                                                                                     \
  {                                                                                                     \
    float ca_[6], cb_[6], ya_[4], yb_[4];                                                               \
    ca_[0] = W4_R(0, B0); ca_[1] = W4_R(1, B0); ca_[2] = W4_R(2, B0); ca_[3] = W4_R(3, B0); ca_[4] = W4_R(4, B0);  \
    ca_[5] = W4_R(5, B0);                                                                               \
    cb_[0] = W4_R(0, B0 + 1); cb_[1] = W4_R(1, B0 + 1); cb_[2] = W4_R(2, B0 + 1); cb_[3] = W4_R(3, B0 + 1);        \
    cb_[4] = W4_R(4, B0 + 1); cb_[5] = W4_R(5, B0 + 1);                                                 \
    w4_landed6(ca_, cb_);                                                                               \
    w4_at(ca_, ya_);                                                                                    \
    w4_at(cb_, yb_);                                                                                    \
    _Pragma("unroll") for (int oa = 0; oa < 4; ++oa) { ycol[oa][B0] = ya_[oa]; ycol[oa][B0 + 1] = yb_[oa]; } \
  }
      W4_COL2(0)
      W4_COL2(2)
#undef W4_COL2
#undef W4_R
#pragma unroll
      for (int oa = 0; oa < 4; ++oa)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
          const float v = fmaxf(__builtin_fmaf(ycol[oa][ob], sc, sh) + rv[mt][oa][ob], act_lo);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, vo[mt], oa * rowpitch + ob * colpitch, 0);
        }
    }
    asm volatile("" ::: "memory");
    W4_CLK()    /* column pass done, stores issued */
    __builtin_amdgcn_s_barrier();      // the exchange buffer is free again (next item's halo)
    asm volatile("" ::: "memory");
    W4_CLK()    /* item end */
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
__global__ __launch_bounds__(W4_NTH, 1) void conv_wino4r_kernel(ConvArgs a) { w4r_body<ABL>(a); }

bool egn_conv_wino4r_applies(const ConvArgs& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 16 == 0 && a.cs_in == a.Cin &&
         a.Cout % W4_CO == 0 && a.cs_out == a.Cout && !a.out_nchw && a.Ho % 16 == 0 && a.Wo % 32 == 0 &&
         !(a.act & EGN_ACT_RES_AFTER) &&
         ((a.act & EGN_ACT_MASK) == EGN_ACT_NONE || (a.act & EGN_ACT_MASK) == EGN_ACT_RELU);
}
size_t egn_conv_wino4r_lds_bytes() { return W4R_LDS + 12 * 96 * 8; }      // (+ the stamp area of the ABL & 64 build)

#ifdef EGN_PROBES
static unsigned w4r_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

template <int ABL>
static int wino4r_launch(ConvArgs a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];
  void (*kern)(ConvArgs) = &conv_wino4r_kernel<ABL>;
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 512));
  }
  const int nct = a.Cout / W4_CO;
  const int nreg = a.tiles_x * a.tiles_y * a.N;
  const int imode = w4_item_mode(nct);
  const int nwork = w4_item_count(imode, nreg, nct, 1);
  if ((unsigned long long)nwork * (unsigned)(8 * nct) >= 0x100000000ull ||
      (unsigned long long)(nreg + 8) * (unsigned)(a.tiles_x * a.tiles_y) >= 0x100000000ull)
    return EGN_E_BADARG;
  a.mg_nct = imode == 1 ? 0u : w4r_magic(imode == 2 ? nct / 8 : nct);
  a.mg_txy = w4r_magic(a.tiles_x * a.tiles_y);
  a.mg_tx = w4r_magic(a.tiles_x);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  int cap = cus / (8 * nct) * (8 * nct);
  if (cap <= 0) cap = 8 * nct;
  const int grid = nwork < cap ? nwork : cap;               // one block per CU, whole XCD rounds
  hipLaunchKernelGGL(kern, dim3(grid), dim3(W4_NTH), lds, stream, a);
  return (int)hipGetLastError();
}
#endif
// Probe builds only: measured SLOWER than conv_wino4_kernel (profiles/r6_wino4r_probe.txt: 54.1 against 50.4 us on
// 48 -> 48 @ 64 x 64, 216.7 against 192.7 on 256 -> 48) -- one m-tile per wave means one filter value per MFMA instead of
// one per two: twice the filter load INSTRUCTIONS (12 per wave and stage), ~780 cycles per stage, more than the single
// exchange round gives back.  The product library does not compile it.
int egn_conv_launch_wino4r(ConvArgs a, size_t lds, int abl, hipStream_t stream) {
#ifdef EGN_PROBES
  if (!egn_conv_wino4r_applies(a)) return EGN_E_BADARG;
  switch (abl) {
    case 0: return wino4r_launch<0>(a, lds, stream);
    case 64: return wino4r_launch<64>(a, lds, stream);
    default: return EGN_E_BADARG;
  }
#else
  (void)a; (void)lds; (void)abl; (void)stream;
  return EGN_E_BADARG;
#endif
}
