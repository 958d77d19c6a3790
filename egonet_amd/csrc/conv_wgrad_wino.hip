// conv_wgrad_wino.hip -- weight gradient of a 3x3 / stride 1 / pad 1 convolution in WINOGRAD F(2x2,3x3)
// form on the fp32 matrix pipe.
//
// Reference: the backward of nn.Conv2d inside the training hot loop (libs/trainer/trainer.py:183-209,
// loss.backward()) for the BasicBlock convs of libs/model/heatmapModel/hrnet.py:68-92 -- 232 of the 306
// weight gradients of HRNet-W48, 5.4 GFLOP each at 32 crops, the largest block of the training step
// (profiles/: 16.5 ms of 67 ms of kernels with the direct kernel of conv_wgrad.hip at 52 % of the fp32 peak).
//
// Forward (conv_wino.hip):  Y_t = A^T [ sum_ci U (.) V_t ] A,  U = G g G^T,  V_t = B^T d_t B  per 2x2 output
// tile t with its 4x4 input patch d_t.  Hence
//     dU_f[co][ci] = sum_t  dM_f[t][co] * V_f[t][ci],    dM_t = A dY_t A^T  (4x4 from the 2x2 tile of dy),
//     dg = G^T dU G                                       (3x3 from 4x4)
// -- 16 multiplies per tile and (ci, co) instead of 36: 2.25x fewer MFMAs than the direct sum, at the
// price of +/- transforms that are lane-local register arithmetic here (nothing transformed touches
// LDS or HBM, the kernel reads x and dy once like the direct one).
//
//   * GEMM view per frequency f = (i, j): M = 48 output channels, N = 48 input channels, K = tiles.
//     v_mfma_f32_16x16x4_f32 with k = 4 TILES per instruction: lane (li = l & 15, kq = l >> 4) holds tile
//     kq of the K step and the channels {li, li + 16, li + 32} of both operands; element ja of dM and jb of
//     V feed MFMA (ja, jb), whose 16x16 result covers rows co = 16 ja + m, columns ci = 16 jb + n.
//   * block = 8 waves, wave (i, jb) owns frequency row i (0..3) and the columns j in {2jb, 2jb+1}: 2 x 9
//     accumulator tiles = 72 registers, two waves per SIMD with room to keep two K steps in flight.
//     Row i of B^T touches two patch rows, columns {2jb, 2jb+1} together three, so a wave reads a 2 x 3
//     sub-patch (18 dwords) + the 2 x 2 dy tile (12 dwords) per K step and issues 18 MFMAs.  Which rows /
//     columns is a per-wave LDS offset, the +/- pattern one sign per direction: all waves run the same code.
//   * LDS stage = the halo image of x and the dy tile, pixel major with 56 dwords per pixel (48 channels
//     + 8 pad): the four tiles of a K step are two pixels = 112 = 48 mod 64 dwords apart, so the 64 lanes
//     of a ds_read_b32 fall into four disjoint 16-bank windows -- conflict free.  Filled by LDS-DMA
//     (buffer_load_dwordx4 ... lds; zero padding = out-of-range lanes), double buffered, one barrier per
//     stage, the next stage's DMA issued behind the stage's first K step.
//   * epilogue: dg = G^T dU G.  A wave folds its two columns into the two distinct column values the
//     three tap columns need, parks them in LDS (one round, 147 KB), and the (tap, element) sums over the
//     eight waves -- fixed order, coefficients G[i][tap row] = +-{0, 1/2, 1} -- are spread over the waves;
//     the block writes ONE partial slab in MFMA fragment order (1 KB contiguous per wave store) and
//     conv_wgrad.hip's reduction kernel sums the slabs of the K splits (deterministic, no atomics).
//
// Numerics: exact fp32 products, fp32 accumulation over the tiles; the transforms add a few ulp and the
// G^T . G step cancels terms of the size of the largest tap (not bit-identical to the direct kernel;
// tests/test_gpu_train_ops.py pins both against float64).
#include <algorithm>
#include <cstdlib>

#include "egn_internal.h"
#include "conv_common.h"
#include "conv_wgrad.h"

typedef __attribute__((address_space(3))) void* lds_ptr_wgw_t;

template <int TW_, int TNB_>
struct WgwGeom {
  static constexpr int TH = 8, TW = TW_, TNB = TNB_;
  static constexpr int HW = TW + 2, HPI = 10 * HW;   // halo columns / halo pixels per image
  static constexpr int PXQ = 14, PXD = 56;           // 16-byte slots / dwords per pixel (12 channel quads + 2 pad)
  static constexpr int HPX = TNB * HPI;              // halo pixels
  static constexpr int DPI = TH * TW;                // dy pixels per image
  static constexpr int DPX = TNB * DPI;
  static constexpr int NHI = (HPX * PXQ + 63) / 64;  // DMA instructions (64 slots each) of the halo image
  static constexpr int NDI = DPX * PXQ / 64;         // ... of the dy tile
  static constexpr int NI = NHI + NDI;
  static constexpr int IT = (NI + 7) / 8;            // per wave
  static constexpr int STAGE = NI * 64;              // 16-byte slots per stage
  static_assert(DPX == 128 && DPX * PXQ % 64 == 0, "32 Winograd tiles = 8 K steps per stage");
  static_assert(2 * STAGE * 16 <= EGN_WGW_LDS_BYTES, "two stages fit");
};

__device__ __forceinline__ void wgw_dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc)
               : "m0");
}

// The eight K steps of a stage for one wave.  Its frequency row i and column pair jb enter only through
// WHICH patch rows / columns it reads (uniform LDS offsets) and three signs:
//   x rows (a, b):    i = 0: (0, 2)  1: (1, 2)  2: (2, 1)  3: (3, 1);   T = a + sr b,  sr = -1, +1, -1, -1
//                     = rows 0, 1, 2, -3 of B^T d
//   x columns (a, b, c) = (0, 2, 1) [jb = 0] or (3, 1, 2) [jb = 1];     V = (Ta - Tb, Tc + sc Tb), sc = +-1
//                     = columns (0, 1) or (-3, 2)
//   dy rows (a, b):   i = 0, 1, 2: (0, 1), i = 3: (1, 0);               R = a + tr b,  tr = 0, +1, -1, 0
//                     = rows 0, 1, 2, -3 of A dY
//   dy columns (a, b) = (0, 1) or (1, 0);                               M = (Ra, Rb + sc Ra)
// Both transforms negate frequency 3, so the products are unchanged.  The wave's accumulator index fj
// therefore means frequency column (0, 1) for jb = 0 and (3, 2) for jb = 1.
struct WgwSel {
  int hro[2], hco[3];   // dword offsets of the patch rows (a, b) / columns (a, b, c)
  int ero[2], eco[2];   // ... of the dy rows / columns (a, b)
  float sr, tr, sc;
};

// One LDS dword at a VGPR byte address + compile-time byte offset, as raw ISA.  Through plain loads the
// compiler pairs the reads into ds_read2_b32 (8-bit offsets) and pays one v_add_u32 per pair for the stage /
// K-step part of the address: 156 address adds per stage of 144 MFMAs -- and beside fp32 MFMAs every VALU
// instruction costs matrix-pipe time (profiles/r3_mfma_tax.txt) while an LDS read costs nothing.  The 16-bit
// immediate of ds_read_b32 holds every offset of a stage.  The reads are asynchronous for the compiler: the
// values are tied to the `s_waitcnt lgkmcnt(0)` below (wgw_landed) before anything uses them.
template <int OFF>
__device__ __forceinline__ float wgw_lds(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b32 immediate");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void wgw_landed(float (&a)[15]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                 "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]));
}

template <int TW, int TNB, int ABL, int S>
__device__ __forceinline__ void wgw_kstep(const unsigned (&ax)[2][3], const unsigned (&ae)[2][2], const WgwSel& w,
                                          int lane, int tile, f32x4 (&acc)[2][3][3]) {
  using G = WgwGeom<TW, TNB>;
  // K step S: tile row S >> 1; the left / right half of the 16-wide tile, or image S & 1 of the pair
  constexpr int ho = ((TNB == 2 ? (S & 1) * G::HPI : 8 * (S & 1)) * G::PXD + 2 * (S >> 1) * G::HW * G::PXD) * 4;
  constexpr int eo = ((TNB == 2 ? (S & 1) * G::DPI : 8 * (S & 1)) * G::PXD + 2 * (S >> 1) * TW * G::PXD) * 4;
  float v[2][15];     // [0]: d[r][c][j] for (r, c, j) = 0..14, [1]: d 15..17 then e[r][c][j] 0..11
  if constexpr ((ABL & 2) != 0) {
#pragma unroll
    for (int k = 0; k < 30; ++k) v[k / 15][k % 15] = (float)(lane + k + tile + S);
  } else {
#define WGW_RD(K) v[(K) / 15][(K) % 15] = (K) < 18 ? wgw_lds<ho + 64 * ((K) % 3)>(ax[((K) / 9) % 2][(((K) / 3) % 3)]) \
                                                      : wgw_lds<eo + 64 * ((K) % 3)>(ae[(((K)-18) / 6) % 2][((((K)-18) / 3) % 2)]);
    WGW_RD(0) WGW_RD(1) WGW_RD(2) WGW_RD(3) WGW_RD(4) WGW_RD(5) WGW_RD(6) WGW_RD(7) WGW_RD(8) WGW_RD(9)
    WGW_RD(10) WGW_RD(11) WGW_RD(12) WGW_RD(13) WGW_RD(14) WGW_RD(15) WGW_RD(16) WGW_RD(17) WGW_RD(18) WGW_RD(19)
    WGW_RD(20) WGW_RD(21) WGW_RD(22) WGW_RD(23) WGW_RD(24) WGW_RD(25) WGW_RD(26) WGW_RD(27) WGW_RD(28) WGW_RD(29)
#undef WGW_RD
    wgw_landed(v[0]);
    wgw_landed(v[1]);
  }
  float d[2][3][3], e[2][2][3];
#pragma unroll
  for (int k = 0; k < 18; ++k) d[k / 9][(k / 3) % 3][k % 3] = v[k / 15][k % 15];
#pragma unroll
  for (int k = 0; k < 12; ++k) e[k / 6][(k / 3) % 2][k % 3] = v[(k + 18) / 15][(k + 18) % 15];
  float T[3][3], V[2][3], R[2][3], M[2][3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[c][j] = __builtin_fmaf(w.sr, d[1][c][j], d[0][c][j]);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    V[0][j] = T[0][j] - T[1][j];
    V[1][j] = __builtin_fmaf(w.sc, T[1][j], T[2][j]);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[c][j] = __builtin_fmaf(w.tr, e[1][c][j], e[0][c][j]);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    M[0][j] = R[0][j];
    M[1][j] = __builtin_fmaf(w.sc, R[0][j], R[1][j]);
  }
#pragma unroll
  for (int fj = 0; fj < 2; ++fj)
#pragma unroll
    for (int ja = 0; ja < 3; ++ja)
#pragma unroll
      for (int jc = 0; jc < 3; ++jc)
        acc[fj][ja][jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(M[fj][ja], V[fj][jc], acc[fj][ja][jc], 0, 0, 0);
}

// ABL != 0: timing ablations (wrong results by construction; EGN_WGW_ABL, tools/wgrad_probe.py only):
//   bit 0 no DMA after the prologue, bit 1 no LDS reads / transforms, bit 2 no cross-wave sum / stores
template <int TW, int TNB, int ABL = 0>
__global__ __launch_bounds__(512, 1) void conv_wgrad_wino_kernel(WgradArgs a) {
  using G = WgwGeom<TW, TNB>;
  constexpr int IT = G::IT;
  extern __shared__ float4 wgw_smem[];
  float* sm = reinterpret_cast<float*>(wgw_smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = wave >> 1, jb = wave & 1;   // frequency row, column pair
  const int li = lane & 15, kq = lane >> 4;
  const int cot = blockIdx.x % a.co_tiles, cit = blockIdx.x / a.co_tiles;
  const int co0 = cot * 48, ci0 = cit * 48;

  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const unsigned long long daddr = reinterpret_cast<unsigned long long>(a.dy);
  const u32x4 rx = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu,
                    (unsigned)((size_t)a.N * a.H * a.W * a.Cin * 4), 0x00020000u};
  const u32x4 rd = {(unsigned)daddr, (unsigned)(daddr >> 32) & 0xffffu,
                    (unsigned)((size_t)a.N * a.H * a.W * a.Cout * 4), 0x00020000u};
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_wgw_t)wgw_smem;

  // DMA slot of this lane in instruction i = wave + 8k: image dn, row dy, column dx of the stage's halo (x) or
  // dy tile and the channel quad.  Per lane and instruction, ONCE: the byte offset relative to the tile's first
  // (image, row - 1, column - 1) and the packed coordinates for the bounds test -- per stage the offset is then
  // `tile base (SGPR) + rel` and the zero-padding test is two subtractions and two ands on guard-bit fields
  // (7 VALU per instruction instead of ~20: beside fp32 MFMAs every VALU instruction costs matrix-pipe time).
  //   mg = 0x808080 | dn << 16 | dy << 8 | dx (a pad slot: all fields 127, never in range)
  //   t1 = mg - lo            field f keeps its guard bit iff v_f >= lo_f
  //   t2 = X - mg             X_f = hi_f + 255: field f = 128 + hi_f - 1 - v_f keeps its guard bit iff v_f < hi_f
  unsigned rel[IT], mg[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = wave + 8 * k;
    int q = 64 * i + lane, img, row, col;
    bool ok;
    if (i < G::NHI) {
      const int px = q / G::PXQ;
      q -= px * G::PXQ;
      img = px / G::HPI;
      const int r = px - img * G::HPI;
      row = r / G::HW;
      col = r - row * G::HW;
      ok = px < G::HPX && q < 12;
    } else {
      q -= G::NHI * 64;
      const int px = q / G::PXQ;
      q -= px * G::PXQ;
      img = px / G::DPI;
      const int r = px - img * G::DPI;
      row = r / TW;
      col = r - row * TW;
      ok = q < 12;
    }
    const int C_ = i < G::NHI ? a.Cin : a.Cout;
    rel[k] = ok ? (unsigned)(((img * a.H + row) * a.W + col) * C_ + 4 * q) * 4u : 0u;
    mg[k] = ok ? (0x808080u | (unsigned)(img << 16) | (unsigned)(row << 8) | (unsigned)col) : 0x00ffffffu;
  }

  const int tiles_xy = a.tiles_x * a.tiles_y;
#define WGW_OFFS(TILE, OUT)                                                                              \
  {                                                                                                      \
    const int tb_ = (TILE) / tiles_xy;                                                                   \
    const int r_ = (TILE)-tb_ * tiles_xy;                                                                \
    const int ty_ = r_ / a.tiles_x, tx_ = r_ - ty_ * a.tiles_x;                                          \
    const int n0_ = tb_ * TNB, y0_ = ty_ * 8, x0_ = tx_ * TW;                                            \
    const unsigned hn_ = (unsigned)min(a.N - n0_, 127) + 255u;                                           \
    /* scalars of the tile: [0] the halo of x (first pixel (y0 - 1, x0 - 1)), [1] the dy tile */         \
    unsigned base_[2], lo_[2], X_[2];                                                                    \
    _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) {                                                   \
      const int ys_ = y0_ - (e_ == 0 ? 1 : 0), xs_ = x0_ - (e_ == 0 ? 1 : 0);                            \
      const int C_ = e_ == 0 ? a.Cin : a.Cout, c0_ = e_ == 0 ? ci0 : co0;                                \
      base_[e_] = (unsigned)(((n0_ * a.H + ys_) * a.W + xs_) * C_ + c0_) * 4u;                           \
      lo_[e_] = (unsigned)max(0, -xs_) | ((unsigned)max(0, -ys_) << 8);                                  \
      X_[e_] = ((unsigned)min(a.W - xs_, 127) + 255u) + (((unsigned)min(a.H - ys_, 127) + 255u) << 8) + (hn_ << 16); \
    }                                                                                                    \
    _Pragma("unroll") for (int k = 0; k < IT; ++k) {                                                     \
      const int e_ = wave + 8 * k < G::NHI ? 0 : 1;   /* wave uniform */                                 \
      const unsigned t_ = (mg[k] - lo_[e_]) & (X_[e_] - mg[k]) & 0x808080u;                              \
      OUT[k] = t_ == 0x808080u ? base_[e_] + rel[k] : EGN_OOB;                                           \
    }                                                                                                    \
  }
#define WGW_ISSUE(K, P, OFF)                                                                             \
  {                                                                                                      \
    const int i_ = wave + 8 * (K);                                                                       \
    if (i_ < G::NI) wgw_dma16(i_ < G::NHI ? rx : rd, lds0 + (unsigned)(((P)*G::STAGE + i_ * 64) * 16), OFF[K]); \
  }

  const int t_begin = blockIdx.y * a.tiles_per_split;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);

  f32x4 acc[2][3][3];  // [fj][ja (co)][jc (ci)]
#pragma unroll
  for (int fj = 0; fj < 2; ++fj)
#pragma unroll
    for (int ja = 0; ja < 3; ++ja)
#pragma unroll
      for (int jc = 0; jc < 3; ++jc) acc[fj][ja][jc] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ABL & 32: phase clocks of wave 0 of every block -> a.part (results are garbage)
  unsigned long long tk[20];
  int ntk = 0;
#define WGW_CLK() { if constexpr ((ABL & 32) != 0) { if (ntk < 20) tk[ntk++] = __builtin_readcyclecounter(); } }
  WGW_CLK()
  unsigned off[IT];
  if (t_begin < t_end) {
    WGW_OFFS(t_begin, off)
#pragma unroll
    for (int k = 0; k < IT; ++k) WGW_ISSUE(k, 0, off)
  }

  // this lane's tile of a K step is tile column kq (+ 4 in the right half); per-wave row / column choice
  const int hb = 2 * kq * G::PXD + li;
  const int db = G::NHI * 64 * 4 + 2 * kq * G::PXD + li;
  WgwSel sel;
  {
    const int pra = fi, prb = fi == 0 ? 2 : (fi == 1 ? 2 : 1);          // (0,2) (1,2) (2,1) (3,1)
    const int pc[3] = {jb == 0 ? 0 : 3, jb == 0 ? 2 : 1, jb == 0 ? 1 : 2};
    sel.hro[0] = pra * G::HW * G::PXD;
    sel.hro[1] = prb * G::HW * G::PXD;
#pragma unroll
    for (int k = 0; k < 3; ++k) sel.hco[k] = pc[k] * G::PXD;
    sel.ero[0] = (fi == 3 ? 1 : 0) * TW * G::PXD; sel.ero[1] = (fi == 3 ? 0 : 1) * TW * G::PXD;
    sel.eco[0] = (jb == 0 ? 0 : 1) * G::PXD;      sel.eco[1] = (jb == 0 ? 1 : 0) * G::PXD;
    sel.sr = fi == 1 ? 1.f : -1.f;
    sel.tr = fi == 1 ? 1.f : (fi == 2 ? -1.f : 0.f);
    sel.sc = jb == 0 ? 1.f : -1.f;
  }
  int par = 0;

  for (int tile = t_begin; tile < t_end; ++tile) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): this wave's pieces of the stage have landed
    __builtin_amdgcn_s_barrier();        // everyone's have; everyone is done reading the other stage
    asm volatile("" ::: "memory");
    WGW_CLK()
    // the ten LDS read bases of the stage (byte addresses: patch rows x columns, dy rows x columns); everything
    // else is an immediate of the ds_read
    unsigned ax[2][3], ae[2][2];
    {
      const unsigned sb = lds0 + (unsigned)(par * (G::STAGE * 16));
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) ax[r][c] = sb + (unsigned)(hb + sel.hro[r] + sel.hco[c]) * 4u;
#pragma unroll
        for (int c = 0; c < 2; ++c) ae[r][c] = sb + (unsigned)(db + sel.ero[r] + sel.eco[c]) * 4u;
      }
    }
    // the first K step goes out before the next stage's DMA is computed and issued: the matrix pipe
    // has work while the offsets are formed (all waves pass the barrier together)
    wgw_kstep<TW, TNB, ABL, 0>(ax, ae, sel, lane, tile, acc);
    if (!(ABL & 1) && tile + 1 < t_end) {
      WGW_OFFS(tile + 1, off)
#pragma unroll
      for (int k = 0; k < IT; ++k) WGW_ISSUE(k, par ^ 1, off)
    }
    asm volatile("" ::: "memory");       // the remaining LDS reads stay below the DMA issue
    wgw_kstep<TW, TNB, ABL, 1>(ax, ae, sel, lane, tile, acc);
    wgw_kstep<TW, TNB, ABL, 2>(ax, ae, sel, lane, tile, acc);
    wgw_kstep<TW, TNB, ABL, 3>(ax, ae, sel, lane, tile, acc);
    wgw_kstep<TW, TNB, ABL, 4>(ax, ae, sel, lane, tile, acc);
    wgw_kstep<TW, TNB, ABL, 5>(ax, ae, sel, lane, tile, acc);
    wgw_kstep<TW, TNB, ABL, 6>(ax, ae, sel, lane, tile, acc);
    wgw_kstep<TW, TNB, ABL, 7>(ax, ae, sel, lane, tile, acc);
    par ^= 1;
    WGW_CLK()
  }
#undef WGW_OFFS
#undef WGW_ISSUE

  // ---- dg = G^T dU G.  Rows of G: (1,0,0) (.5,.5,.5) (.5,-.5,.5) (0,0,1).  Along the columns the wave's
  // (v0, v1) = frequency columns (0, 1) [jb = 0] or (3, 2) [jb = 1] give the two values q0 = v0 + v1/2,
  // q1 = v1/2 that serve tap columns (q0, q1, q1) [jb = 0] or (q1, -q1, q0) [jb = 1]; along the rows wave
  // (i, .) enters tap row ta with G[i][ta].  One LDS round: park q0, q1 as red[w][q][e][lane], then the
  // 81 (tap, e) sums over the waves (fixed order) are spread over the waves.  The block's slab is written
  // in FRAGMENT ORDER [tap][e = ja*3+jc][lane][r] (co = 16ja + 4kq + r, ci = 16jc + li): a wave store is
  // 1 KB contiguous; conv_wgrad.hip's reduction kernel decodes it.
  f32x4* red = reinterpret_cast<f32x4*>(wgw_smem);
  if constexpr ((ABL & 4) != 0) {
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 9; ++e) s += acc[q][e / 3][e % 3];
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) a.part[tid] = s[0];
    return;
  }
  f32x4* slab = reinterpret_cast<f32x4*>(a.part) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (81 * 64) + lane;
  WGW_CLK()
  __syncthreads();   // the stages are consumed
  WGW_CLK()
#pragma unroll
  for (int ja = 0; ja < 3; ++ja)
#pragma unroll
    for (int jc = 0; jc < 3; ++jc) {
      const f32x4 h = 0.5f * acc[1][ja][jc];
      red[((wave * 2 + 0) * 9 + ja * 3 + jc) * 64 + lane] = acc[0][ja][jc] + h;
      red[((wave * 2 + 1) * 9 + ja * 3 + jc) * 64 + lane] = h;
    }
  __syncthreads();
  WGW_CLK()
  // 81 items, wave w takes w, w + 8, ...: two at a time so that the LDS round trips overlap
  auto item_sum = [&](int item) -> f32x4 {
    const int tap = item / 9, e = item - tap * 9;
    const int ta = tap / 3, tb = tap - ta * 3;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // G[i][ta]
      const float g = i == 0 ? (ta == 0 ? 1.f : 0.f) : i == 1 ? .5f : i == 2 ? (ta == 1 ? -.5f : .5f) : (ta == 2 ? 1.f : 0.f);
      if (g != 0.f) {
        const f32x4 v0 = red[(((2 * i + 0) * 2 + (tb == 0 ? 0 : 1)) * 9 + e) * 64 + lane];    // jb = 0: (q0, q1, q1)
        const f32x4 v1 = red[(((2 * i + 1) * 2 + (tb == 2 ? 0 : 1)) * 9 + e) * 64 + lane];    // jb = 1: (q1, -q1, q0)
        s += g * (tb == 1 ? v0 - v1 : v0 + v1);
      }
    }
    return s;
  };
  int item = wave;
  for (; item + 8 < 81; item += 16) {
    const f32x4 s0 = item_sum(item), s1 = item_sum(item + 8);
    slab[item * 64] = s0;
    slab[(item + 8) * 64] = s1;
  }
  if (item < 81) slab[item * 64] = item_sum(item);
  if constexpr ((ABL & 32) != 0) {
    __syncthreads();
    WGW_CLK()
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.part) + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32;
      o[0] = ntk;
      for (int i = 0; i < 20; ++i) o[1 + i] = tk[i];
    }
  }
#undef WGW_CLK
}

template <int TW, int TNB, int ABL = 0>
static int wgw_launch(const WgradArgs& a, hipStream_t stream) {
  auto k = conv_wgrad_wino_kernel<TW, TNB, ABL>;
  static bool raised[EGN_MAX_DEVICES];
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  }
  hipLaunchKernelGGL(k, dim3(a.co_tiles * a.ci_tiles, a.nsplit), dim3(512), EGN_WGW_LDS_BYTES, stream, a);
  return (int)hipGetLastError();
}

int egn_wgrad_wino_launch(const WgradArgs& a, int variant, hipStream_t stream) {
#ifdef EGN_PROBES      // timing-ablation / stamp builds (WRONG RESULTS): probe builds only, tools/wgrad_probe.py / wgw_clk.py
  static const int abl = getenv("EGN_WGW_ABL") ? atoi(getenv("EGN_WGW_ABL")) : 0;
  if (abl && variant != 2) {
    switch (abl) {
      case 1: return wgw_launch<16, 1, 1>(a, stream);
      case 2: return wgw_launch<16, 1, 2>(a, stream);
      case 3: return wgw_launch<16, 1, 3>(a, stream);
      case 4: return wgw_launch<16, 1, 4>(a, stream);
      case 7: return wgw_launch<16, 1, 7>(a, stream);
      case 32: return wgw_launch<16, 1, 32>(a, stream);
      default: break;
    }
  }
#endif
  return variant == 2 ? wgw_launch<8, 2>(a, stream) : wgw_launch<16, 1>(a, stream);
}
