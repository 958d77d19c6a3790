// conv_wgrad.hip -- weight gradient of a convolution / Linear layer on fp32 MFMA.
//
// Reference: the backward pass autograd runs for nn.Conv2d / nn.Linear inside
// the training hot loop, libs/trainer/trainer.py:183-209 (loss.backward()), for
// the layers of libs/model/heatmapModel/hrnet.py and libs/model/FCmodel.py:
//     dW[co][ci][ky][kx] = sum_{n,oy,ox} dy[n][oy][ox][co] * x[n][oy*s+ky-p][ox*s+kx-p][ci]
//
// GEMM view: M = Cout, N = Cin (per tap), K = output pixels.  K is the huge
// dimension (N*Ho*Wo), M x N x taps is small, so the parallelism comes from
// splitting K: grid = (co-tile x ci-tile, split); every block walks over its
// range of pixel tiles, keeps its [CW x IW x taps] accumulators in registers and
// writes ONE partial at the end; a second kernel sums the partials in a fixed
// order (deterministic, no atomics) into torch's [Cout][Cin][KH][KW] layout.
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, k = 4 pixels per instruction).  Both
// operands are NHWC, i.e. channel-contiguous per pixel, so one ds_read_b128 of
// a lane brings 4 consecutive CHANNELS of one pixel.  Lane l = (i = l & 15,
// kq = l >> 4) reads dy[pixel 4s+kq][co 4i..4i+3] and x[shifted pixel][ci 4i..4i+3];
// element ja of the first and jb of the second feed MFMA (ja, jb), whose 16x16
// result covers output rows {4m+ja} x columns {4n+jb}: 16 MFMAs per two LDS
// reads, a 64 x 64 (co x ci) tile per wave and tap.  The filter taps are
// spread over the waves of a block (3x3: nine waves share one staged dy tile
// and one x halo tile), so the tap shift is only an LDS address offset.
#include <algorithm>
#include <cstdlib>

#include "egn_internal.h"
#include "conv_common.h"
#include "conv_wgrad.h"
#include <atomic>
extern std::atomic<long> g_egn_direct_convs;   // program.hip


// J = channels per lane and operand (4: 64-wide tiles, ds_read_b128; 3: 48-wide
// tiles for the 48 / 96 channel branches, 9 instead of 16 MFMAs per K step).
// PF = software pipeline: the next pixel tile's global loads are issued into
// registers before the MFMA loop of the current one (A_IT / B_IT float4 per lane).
//
// KS > 1 = K slices inside the block: KS groups of NTAPW*WM*WN waves share the staged
// tiles, group ks takes the K steps s = ks (mod KS), and the groups' accumulators are
// summed through LDS (fixed order) before the ONE partial of the block is written.
// <3,3,1,1,3,..,4>: three waves own one filter column each (TPW = 3 taps, 27 MFMAs per
// K step and four LDS reads), four K slices -> 12 waves = 3 per SIMD, where the
// 9-waves-one-per-tap layout leaves the SIMDs with 3/2/2/2 waves (130 VGPRs: one block
// per CU) -- the 3x3 layers of the 48/96/192/384-channel branches.
template <int NTAPW, int TPW, int WM, int WN, int J, int A_IT, int B_IT, int KS = 1>
__global__ __launch_bounds__(64 * NTAPW * WM * WN * KS) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int NT = 64 * NTAPW * WM * WN * KS;
  constexpr int NW1 = NTAPW * WM * WN;       // waves of one K slice
  constexpr int WT = 16 * J;                 // channels per wave tile
  constexpr int CW = WT * WM, IW = WT * WN;  // channels per block tile
  constexpr int CW4 = CW / 4, IW4 = IW / 4;
  constexpr int LDA = CW + 4, LDB = IW + 4;  // +4 floats: rows stay 16 B aligned, banks rotate
  extern __shared__ float wsm[];
  float* sA = wsm;                // [TP][LDA]   dy tile
  float* sB = wsm + a.TP * LDA;   // [NHP][LDB]  x halo tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave_all / NW1;
  const int wave = wave_all % NW1;
  const int tapw = wave / (WM * WN);
  const int wm = (wave / WN) % WM;
  const int wn = wave % WN;
  const int li = lane & 15;
  const int kq = lane >> 4;

  const int cot = blockIdx.x % a.co_tiles;
  const int cit = blockIdx.x / a.co_tiles;
  const int co0 = cot * CW, ci0 = cit * IW;

  f32x4 acc[TPW][J][J];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int ja = 0; ja < J; ++ja)
#pragma unroll
      for (int jb = 0; jb < J; ++jb) acc[t][ja][jb] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tap_off[TPW];  // LDS pixel offset of this wave's taps inside the halo tile (-1: no such tap)
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = tapw + t * NTAPW;
    tap_off[t] = tap < a.taps ? (tap / a.KW) * a.HWd + (tap % a.KW) : -1;
  }

  const int t_begin = blockIdx.y * a.tiles_per_split;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
  const int thw_mask = (1 << a.lg_thw) - 1;
  const int tw_mask = a.TW - 1;
  const int hpi = a.HH * a.HWd;  // halo pixels per image
  const int a_elems = a.TP * CW4, b_elems = a.NHP * IW4;

  float4 ra[A_IT], rb[B_IT];

// global -> registers for pixel tile TILE (zeros outside the image / batch / channels)
#define EGN_WG_LOAD(TILE)                                                                               \
  {                                                                                                     \
    const int tx_ = (TILE) % a.tiles_x;                                                                 \
    const int ty_ = ((TILE) / a.tiles_x) % a.tiles_y;                                                   \
    const int tb_ = (TILE) / (a.tiles_x * a.tiles_y);                                                   \
    const int nb_ = tb_ * a.TNB, oy0_ = ty_ * a.TH, ox0_ = tx_ * a.TW;                                  \
    _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                               \
      const int e = tid + it * NT;                                                                      \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
      if (e < a_elems) {                                                                                \
        const int p = e / CW4, c4 = e - p * CW4;                                                        \
        const int b = p >> a.lg_thw, rem = p & thw_mask;                                                \
        const int n = nb_ + b, oy = oy0_ + (rem >> a.lg_tw), ox = ox0_ + (rem & tw_mask);               \
        const int c = co0 + 4 * c4;                                                                     \
        if (n < a.N && oy < a.Ho && ox < a.Wo && c < a.cs_out)                                          \
          v = *reinterpret_cast<const float4*>(a.dy + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.cs_out + c); \
      }                                                                                                 \
      ra[it] = v;                                                                                       \
    }                                                                                                   \
    _Pragma("unroll") for (int it = 0; it < B_IT; ++it) {                                               \
      const int e = tid + it * NT;                                                                      \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
      if (e < b_elems) {                                                                                \
        const int hp = e / IW4, c4 = e - hp * IW4;                                                      \
        const int b = hp / hpi, r = hp - b * hpi;                                                       \
        const int hy = r / a.HWd, hx = r - hy * a.HWd;                                                  \
        const int n = nb_ + b;                                                                          \
        const int iy = oy0_ * a.stride - a.pad + hy, ix = ox0_ * a.stride - a.pad + hx;                 \
        const int c = ci0 + 4 * c4;                                                                     \
        if (n < a.N && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.cs_in)                       \
          v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.cs_in + c); \
      }                                                                                                 \
      rb[it] = v;                                                                                       \
    }                                                                                                   \
  }

  if (t_begin < t_end) EGN_WG_LOAD(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();  // the previous tile's fragments are consumed
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int e = tid + it * NT;
      if (e < a_elems) {
        const int p = e / CW4, c4 = e - p * CW4;
        *reinterpret_cast<float4*>(sA + p * LDA + 4 * c4) = ra[it];
      }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int e = tid + it * NT;
      if (e < b_elems) {
        const int hp = e / IW4, c4 = e - hp * IW4;
        *reinterpret_cast<float4*>(sB + hp * LDB + 4 * c4) = rb[it];
      }
    }
    __syncthreads();
    if (tile + 1 < t_end) EGN_WG_LOAD(tile + 1);  // in flight during the MFMA loop below
    const float* pa = sA + wm * WT + J * li;
    const float* pb = sB + wn * WT + J * li;
    for (int s = ks; s < a.TP / 4; s += KS) {
      const int p = 4 * s + kq;
      const int b = p >> a.lg_thw, rem = p & thw_mask;
      const int hbase = (b * a.HH + (rem >> a.lg_tw) * a.stride) * a.HWd + (rem & tw_mask) * a.stride;
      float av[J];
#pragma unroll
      for (int j = 0; j < J; ++j) av[j] = pa[p * LDA + j];
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (KS > 1 || tap_off[t] >= 0) {   // the K-sliced variant only runs full 3x3 filters
          float bv[J];
#pragma unroll
          for (int j = 0; j < J; ++j) bv[j] = pb[(hbase + tap_off[t]) * LDB + j];
#pragma unroll
          for (int ja = 0; ja < J; ++ja)
#pragma unroll
            for (int jb = 0; jb < J; ++jb)
              acc[t][ja][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ja], bv[jb], acc[t][ja][jb], 0, 0, 0);
        }
      }
    }
  }
#undef EGN_WG_LOAD

  if constexpr (KS > 1) {
    // sum the K slices: slice r parks its accumulators in LDS, slice 0 adds them, r = 1..KS-1
    f32x4* rbuf = reinterpret_cast<f32x4*>(wsm) + (size_t)wave * (TPW * J * J) * 64 + lane;
    for (int r = 1; r < KS; ++r) {
      __syncthreads();
      if (ks == r) {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int ja = 0; ja < J; ++ja)
#pragma unroll
            for (int jb = 0; jb < J; ++jb) rbuf[((t * J + ja) * J + jb) * 64] = acc[t][ja][jb];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int ja = 0; ja < J; ++ja)
#pragma unroll
            for (int jb = 0; jb < J; ++jb) acc[t][ja][jb] += rbuf[((t * J + ja) * J + jb) * 64];
      }
    }
    if (ks != 0) return;
  }

  // partial [split][tap][CoP][CiP]: lane owns rows J*(4kq+r)+ja, columns J*li .. J*li+J-1
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = tapw + t * NTAPW;
    if (tap < a.taps) {
      float* dst = a.part + ((size_t)(blockIdx.y * a.taps + tap) * a.CoP + co0 + wm * WT) * a.CiP + ci0 + wn * WT + J * li;
#pragma unroll
      for (int ja = 0; ja < J; ++ja)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* q = dst + (size_t)(J * (4 * kq + r) + ja) * a.CiP;
#pragma unroll
          for (int jb = 0; jb < J; ++jb) q[jb] = acc[t][ja][jb][r];
        }
    }
  }
}

// dw[co][ci][tap] = sum_s part[s][tap][co][ci]   (fixed order: deterministic)
// The partial slabs are dense [tap][CoP][CiP] (CiP % 4 == 0): block = 32 float4 elements
// x LANES split lanes (32 for many splits, 8 for few), 512 B contiguous per wave load and
// split; lane l sums splits l, l+LANES, ... (eight loads in flight), the lane sums are
// combined in lane order.
template <int LANES>
__global__ __launch_bounds__(32 * LANES) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                  int nsplit, int taps, int Cout, int Cin, int CoP,
                                                                  int CiP, int frag_co_tiles) {
  __shared__ float4 red[LANES][32];
  const int ci4n = CiP >> 2;
  const size_t total4 = (size_t)taps * CoP * ci4n;
  const int el = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const size_t e = (size_t)blockIdx.x * 32 + el;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < total4) {
    const size_t stride4 = total4;
    const float4* q = reinterpret_cast<const float4*>(part) + e;
    for (int k0 = lane; k0 < nsplit; k0 += 8 * LANES) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + LANES * u;
        v[u] = k < nsplit ? q[(size_t)k * stride4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  red[lane][el] = s;
  __syncthreads();
  if (lane == 0 && e < total4) {
    float4 t = red[0][el];
#pragma unroll 4
    for (int r = 1; r < LANES; ++r) { const float4 o = red[r][el]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    if (frag_co_tiles > 0) {
      // conv_wgrad_wino.hip slabs: [co-tile + ci-tile * co_tiles][tap][e = ja*3+jc][lane][r] in MFMA fragment
      // order -- co = 16 ja + 4 (lane >> 4) + r, ci = 16 jc + (lane & 15) inside the 48 x 48 tile
      const int ln = (int)(e & 63), ee = (int)((e >> 6) % 9), tap = (int)((e / 576) % 9), blk = (int)(e / 5184);
      const int co = (blk % frag_co_tiles) * 48 + 16 * (ee / 3) + 4 * (ln >> 4);
      const int ci = (blk / frag_co_tiles) * 48 + 16 * (ee % 3) + (ln & 15);
      float* d = dw + ((size_t)co * Cin + ci) * taps + tap;
      const size_t rs = (size_t)Cin * taps;
      d[0] = t.x; d[rs] = t.y; d[2 * rs] = t.z; d[3 * rs] = t.w;
      return;
    }
    const int ci = (int)(e % ci4n) * 4;
    const int co = (int)((e / ci4n) % CoP);
    const int tap = (int)(e / ((size_t)ci4n * CoP));
    if (co < Cout) {
      float* d = dw + ((size_t)co * Cin + ci) * taps + tap;
      if (ci + 0 < Cin) d[0] = t.x;
      if (ci + 1 < Cin) d[(size_t)taps] = t.y;
      if (ci + 2 < Cin) d[(size_t)2 * taps] = t.z;
      if (ci + 3 < Cin) d[(size_t)3 * taps] = t.w;
    }
  }
}

static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

constexpr int EGN_WGRAD_MAX_SPLITS = 512;

struct WgradVariant {
  int ntapw, tpw, wm, wn;
  int j, a_it, b_it;  // channels per lane, register-staging depth (float4 per lane) of the dy / x tiles
  int ks;             // K slices inside the block (0 = 1)
  int wino;           // conv_wgrad_wino.hip variant (0 = direct kernels of this file)
};

static int pad_to(int v, int m) { return (v + m - 1) / m * m; }
// 48-wide (J = 3) or 64-wide (J = 4) wave tiles: whichever pads Cout x Cin less
static int pick_j(int cout, int cin, int wm, int wn, bool prefer3 = false) {
  const long c3 = (long)pad_to(cout, 48 * wm) * pad_to(cin, 48 * wn);
  const long c4 = (long)pad_to(cout, 64 * wm) * pad_to(cin, 64 * wn);
  return (c3 < c4 || (prefer3 && c3 == c4)) ? 3 : 4;
}

static int wgrad_plan(WgradArgs& a, WgradVariant& v, size_t& lds) {
  if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.Cin <= 0 || a.Cout <= 0 || a.KH <= 0 || a.KW <= 0 || a.stride <= 0 ||
      a.pad < 0 || a.cs_in % 4 || a.cs_out % 4 || a.cs_in < a.Cin || a.cs_out < a.Cout)
    return EGN_E_BADARG;
  a.Ho = (a.H + 2 * a.pad - a.KH) / a.stride + 1;
  a.Wo = (a.W + 2 * a.pad - a.KW) / a.stride + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return EGN_E_BADARG;
  a.taps = a.KH * a.KW;
  // 3x3 / stride 1 / pad 1 layers with 48-multiple channel counts (the HRNet-W48 branches): the Winograd
  // form, 2.25x fewer MFMAs (conv_wgrad_wino.hip).  EGN_WGRAD_WINO=0: direct kernels only.
  static const bool wino_on = !(getenv("EGN_WGRAD_WINO") && !atoi(getenv("EGN_WGRAD_WINO")));
  if (wino_on && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 48 == 0 && a.Cout % 48 == 0 &&
      a.cs_in == a.Cin && a.cs_out == a.Cout && (size_t)a.N * a.H * a.W * std::max(a.Cin, a.Cout) * 4 < (1ull << 31)) {
    v = WgradVariant{};
    v.wino = a.W <= 8 ? 2 : 1;
    a.TH = 8; a.TW = v.wino == 2 ? 8 : 16; a.TNB = v.wino == 2 ? 2 : 1;
    a.tiles_x = (a.W + a.TW - 1) / a.TW;
    a.tiles_y = (a.H + 7) / 8;
    a.ntiles = a.tiles_x * a.tiles_y * ((a.N + a.TNB - 1) / a.TNB);
    a.co_tiles = a.Cout / 48; a.ci_tiles = a.Cin / 48;
    a.CoP = a.Cout; a.CiP = a.Cin;
    const int base = a.co_tiles * a.ci_tiles;
    int want = (256 + base - 1) / base;        // one 8-wave block per CU
    want = std::max(1, std::min(std::min(want, a.ntiles), EGN_WGRAD_MAX_SPLITS));
    a.tiles_per_split = (a.ntiles + want - 1) / want;
    a.nsplit = (a.ntiles + a.tiles_per_split - 1) / a.tiles_per_split;
    lds = EGN_WGW_LDS_BYTES;
    return 0;
  }
  if (a.taps == 1) {
    v = (a.Cout > 64 && a.Cin > 64) ? WgradVariant{1, 1, 2, 2, 4, 4, 4}
        : (a.Cout > 64)             ? WgradVariant{1, 1, 2, 1, 4, 8, 4}
        : (a.Cin > 64)              ? WgradVariant{1, 1, 1, 2, 4, 4, 8}
                                    : WgradVariant{1, 1, 1, 1, pick_j(a.Cout, a.Cin, 1, 1), 8, 8};
  } else if (a.taps <= 9) {
    static const bool sliced = !(getenv("EGN_WGRAD_KSLICE") && !atoi(getenv("EGN_WGRAD_KSLICE")));
    const int j = pick_j(a.Cout, a.Cin, 1, 1, sliced && a.KH == 3 && a.KW == 3);
    if (sliced && a.KH == 3 && a.KW == 3 && j == 3)
      v = WgradVariant{3, 3, 1, 1, 3, 2, 5, 4};                        // one wave per filter column x 4 K slices
    else
      v = WgradVariant{9, 1, 1, 1, j, 2, 5};                           // waves whose tap does not exist idle
  } else if (a.taps <= 16) {
    v = WgradVariant{8, 2, 1, 1, pick_j(a.Cout, a.Cin, 1, 1), 2, 5};
  } else {
    return EGN_E_BADARG;
  }
  const int NT = 64 * v.ntapw * v.wm * v.wn * std::max(1, v.ks);
  const int CW = 16 * v.j * v.wm, IW = 16 * v.j * v.wn;
  a.co_tiles = (a.Cout + CW - 1) / CW;
  a.ci_tiles = (a.Cin + IW - 1) / IW;
  a.CoP = a.co_tiles * CW;
  a.CiP = a.ci_tiles * IW;
  // pixels per tile: 64, or 128 (8 x 16) for the 48-wide 3x3 variant, whose two tiles still
  // fit twice per CU (65.7 KB): half as many barriers per MFMA
  static const bool big_tiles = !(getenv("EGN_WGRAD_TP64") && atoi(getenv("EGN_WGRAD_TP64")));
  const bool tp128 = big_tiles && (v.ntapw == 9 || v.ks == 4) && v.j == 3 && a.stride == 1;
  if (tp128 && v.ks == 4) v.b_it = 3;      // 768 threads: 128 x 12 and 180 x 12 float4 in 2 + 3 rounds
  else if (tp128) v.a_it = 3;
  const int tp_target = tp128 ? 128 : 64;
  a.TW = std::min(pow2_ceil(a.Wo), tp128 ? 16 : 8);
  a.TH = std::min(pow2_ceil(a.Ho), 8);
  a.TNB = std::max(1, tp_target / (a.TH * a.TW));
  const size_t budget = tp128 ? 66 * 1024 : 64 * 1024;
  for (;;) {
    if (a.TH * a.TW * a.TNB < 4) a.TNB = 4 / (a.TH * a.TW);
    a.HH = (a.TH - 1) * a.stride + a.KH;
    a.HWd = (a.TW - 1) * a.stride + a.KW;
    a.TP = a.TNB * a.TH * a.TW;
    a.NHP = a.TNB * a.HH * a.HWd;
    lds = ((size_t)a.TP * (CW + 4) + (size_t)a.NHP * (IW + 4)) * sizeof(float);
    const bool regs_ok = a.TP * (CW / 4) <= v.a_it * NT && a.NHP * (IW / 4) <= v.b_it * NT;
    if (lds <= budget && regs_ok) break;
    if (a.TNB > 1 && a.TNB * a.TH * a.TW > 4) a.TNB /= 2;
    else if (a.TH > 1 && a.TH >= a.TW) a.TH /= 2;
    else if (a.TW > 1) a.TW /= 2;
    else if (lds <= 150 * 1024 && regs_ok) break;
    else return EGN_E_BADARG;
  }
  if (v.ks > 1)  // the K slices meet in LDS: ntapw waves x tpw*j*j accumulators x 64 lanes x 16 B
    lds = std::max(lds, (size_t)v.ntapw * v.wm * v.wn * v.tpw * v.j * v.j * 64 * 16);
  a.lg_tw = ilog2_exact(a.TW);
  a.lg_thw = ilog2_exact(a.TH * a.TW);
  a.tiles_x = (a.Wo + a.TW - 1) / a.TW;
  a.tiles_y = (a.Ho + a.TH - 1) / a.TH;
  const int tiles_b = (a.N + a.TNB - 1) / a.TNB;
  a.ntiles = a.tiles_x * a.tiles_y * tiles_b;
  const int base = a.co_tiles * a.ci_tiles;
  // blocks the launch should have: ~2 per CU; more splits = more partial traffic
  // (each split writes taps*CoP*CiP floats), fewer = idle CUs
  static const int target = [] {
    const char* e = getenv("EGN_WGRAD_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 256;
  }();
  int want = (target + base - 1) / base;
  want = std::max(1, std::min(std::min(want, a.ntiles), EGN_WGRAD_MAX_SPLITS));
  a.tiles_per_split = (a.ntiles + want - 1) / want;
  a.nsplit = (a.ntiles + a.tiles_per_split - 1) / a.tiles_per_split;
  return 0;
}

template <int NTAPW, int TPW, int WM, int WN, int J, int A_IT, int B_IT, int KS = 1>
static int wgrad_launch(const WgradArgs& a, size_t lds, hipStream_t stream) {
  auto k = conv_wgrad_kernel<NTAPW, TPW, WM, WN, J, A_IT, B_IT, KS>;
  static bool raised[EGN_MAX_DEVICES];  // per instantiation and device
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  }
  hipLaunchKernelGGL(k, dim3(a.co_tiles * a.ci_tiles, a.nsplit), dim3(64 * NTAPW * WM * WN * KS), lds, stream, a);
  return (int)hipGetLastError();
}

extern "C" long egn_conv2d_wgrad_ws_bytes(int N, int H, int W, int Cin, int cs_in, int Cout, int cs_out, int KH,
                                          int KW, int stride, int pad) {
  WgradArgs a = {};
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.cs_in = cs_in; a.Cout = Cout; a.cs_out = cs_out;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  WgradVariant v;
  size_t lds;
  if (wgrad_plan(a, v, lds) != 0) return -1;
  return (long)((size_t)a.nsplit * a.taps * a.CoP * a.CiP * sizeof(float));
}

extern "C" int egn_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin,
                                    int cs_in, int Cout, int cs_out, int KH, int KW, int stride, int pad, void* ws,
                                    long ws_bytes, void* stream) {
  if (!x || !dy || !dw || !ws) return EGN_E_BADARG;
  g_egn_direct_convs.fetch_add(1, std::memory_order_relaxed);
  WgradArgs a = {};
  a.x = x; a.dy = dy; a.part = (float*)ws;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.cs_in = cs_in; a.Cout = Cout; a.cs_out = cs_out;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  WgradVariant v;
  size_t lds;
  int rc = wgrad_plan(a, v, lds);
  if (rc != 0) return rc;
  if ((size_t)ws_bytes < (size_t)a.nsplit * a.taps * a.CoP * a.CiP * sizeof(float)) return EGN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (v.wino) rc = egn_wgrad_wino_launch(a, v.wino, st);
  else if (v.ks == 4) rc = v.b_it == 3 ? wgrad_launch<3, 3, 1, 1, 3, 2, 3, 4>(a, lds, st) : wgrad_launch<3, 3, 1, 1, 3, 2, 5, 4>(a, lds, st);
  else if (v.ntapw == 9 && v.j == 3) rc = v.a_it == 3 ? wgrad_launch<9, 1, 1, 1, 3, 3, 5>(a, lds, st) : wgrad_launch<9, 1, 1, 1, 3, 2, 5>(a, lds, st);
  else if (v.ntapw == 9) rc = wgrad_launch<9, 1, 1, 1, 4, 2, 5>(a, lds, st);
  else if (v.ntapw == 8) rc = v.j == 3 ? wgrad_launch<8, 2, 1, 1, 3, 2, 5>(a, lds, st) : wgrad_launch<8, 2, 1, 1, 4, 2, 5>(a, lds, st);
  else if (v.wm == 2 && v.wn == 2) rc = wgrad_launch<1, 1, 2, 2, 4, 4, 4>(a, lds, st);
  else if (v.wm == 2) rc = wgrad_launch<1, 1, 2, 1, 4, 8, 4>(a, lds, st);
  else if (v.wn == 2) rc = wgrad_launch<1, 1, 1, 2, 4, 4, 8>(a, lds, st);
  else rc = v.j == 3 ? wgrad_launch<1, 1, 1, 1, 3, 8, 8>(a, lds, st) : wgrad_launch<1, 1, 1, 1, 4, 8, 8>(a, lds, st);
  if (rc != 0) return rc;
  const size_t total4 = (size_t)a.taps * a.CoP * (a.CiP / 4);
  if (a.nsplit > 64)
    hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3((unsigned)((total4 + 31) / 32)), dim3(1024), 0, st, a.part, dw,
                       a.nsplit, a.taps, Cout, Cin, a.CoP, a.CiP, v.wino ? a.co_tiles : 0);
  else if (a.nsplit > 4)
    hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)((total4 + 31) / 32)), dim3(256), 0, st, a.part, dw,
                       a.nsplit, a.taps, Cout, Cin, a.CoP, a.CiP, v.wino ? a.co_tiles : 0);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<2>, dim3((unsigned)((total4 + 31) / 32)), dim3(64), 0, st, a.part, dw,
                       a.nsplit, a.taps, Cout, Cin, a.CoP, a.CiP, v.wino ? a.co_tiles : 0);
  return (int)hipGetLastError();
}
