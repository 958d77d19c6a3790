// conv_mfma.hip -- fused convolution for gfx950 as an fp32-MFMA implicit GEMM.
//
//   y = act( conv(x, w) * scale + shift (+ res) )
//
// GEMM view: M = output pixels of a (TNB x TH x TW) tile, N = output channels,
// K = (tap, input channel).  K is walked in chunks of 16 input channels; for a
// chunk the input halo tile of the block is staged ONCE in LDS and re-used by
// every filter tap (the 3x3 stencil re-use happens in LDS, not in HBM), the
// weights of the chunk are staged per group of taps ("stage").
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain).  Lane l
// supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it owns
// C[4*(l>>4) + r][l&15], r = 0..3.  Each lane-quad kq = l>>4 is given 4
// consecutive input channels of the chunk, so one ds_read_b128 feeds 4 MFMA
// k-steps (the K order inside a chunk is a permutation of the reference's
// summation order, which fp32 parity allows).
//
// Pipeline: the global loads of stage s+1 (<= 8 + 8 dwordx4 per lane) are
// issued into registers right before the MFMA loop of stage s and committed to
// LDS after it, so HBM/L2 latency hides under ~14k cycles of MFMA work; with
// ~50 KB of LDS per block three blocks share a CU and cover each other's
// barrier / ds_write phases.
//
// LDS layout (float4 granules, 16 B):
//   sA[q][pos]   q = channel quad 0..3, pos = halo pixel ^ (q<<1)   (XOR swizzle
//                keeps both the ds_write_b128 fill and the ds_read_b128 fragment
//                reads on distinct 16-B slots), plane stride npixp (mult. of 16)
//   sB[t][q][co] weights of tap t, channel quad q, output channel co (TN wide)
// Epilogue (NHWC): accumulators go through LDS (sC[row][col], row stride
// TNW+4) so that every lane stores / reads residuals as 16-B float4 along the
// channel axis; sPix[m] holds the output pixel index of tile row m.
#include "egn_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// A_IT / B_IT (template): max dwordx4 loads per lane for the halo tile of a chunk / the weights of a stage

__device__ __forceinline__ float egn_act(float v, int act) {
  switch (act) {
    case EGN_ACT_RELU: return fmaxf(v, 0.0f);
    case EGN_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case EGN_ACT_LEAKY: return v > 0.0f ? v : 0.01f * v;
    default: return v;
  }
}

template <int WM, int WN, int MT, int NT, int A_IT, int B_IT>
__global__ __launch_bounds__(256, 3) void conv_mfma_kernel(ConvArgs a) {
  static_assert(WM * WN == 4, "256-thread workgroups");
  constexpr int NTHREADS = 256;
  constexpr int TN = WN * NT * 16;
  constexpr int TNW = NT * 16;      // columns per wave
  constexpr int TM = WM * MT * 16;  // rows per block
  constexpr int CKQ = EGN_CKQ;

  extern __shared__ float4 smem[];
  float4* sA = smem;
  float4* sB = smem + CKQ * a.npixp;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 15;
  const int kq = lane >> 4;

  const int tile = blockIdx.x;
  const int tx = tile % a.tiles_x;
  const int ty = (tile / a.tiles_x) % a.tiles_y;
  const int tb = tile / (a.tiles_x * a.tiles_y);
  const int n_base = tb * a.TNB;
  const int oy0 = ty * a.TH;
  const int ox0 = tx * a.TW;
  const int n0 = blockIdx.y * TN;
  const int tile_px = a.TH * a.TW;

  // ---- staged loads go through raw buffer resources: 32-bit per-lane byte
  // offsets (no 64-bit address VGPRs), a wave-uniform SGPR offset selects the
  // chunk / stage, and out-of-range offsets return 0 -- which IS the zero
  // padding of the convolution.
  constexpr unsigned OOB = 0xF0000000u;  // > any tensor size accepted by the planner
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (unsigned)((size_t)a.N * a.H * a.W * a.cs_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.w), 0, (unsigned)((size_t)a.nchunk * a.taps * CKQ * a.CoutP * 16), 0x00020000);
  // halo tile: element e = tid + it*256 -> quad q = tid & 3, pixel p = (tid>>2) + it*64
  const int q = tid & 3;
  const int p0 = tid >> 2;
  unsigned aoff[A_IT];  // byte offset of (pixel p, quad q) in x; OOB = zero padding / beyond the tile
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int p = p0 + it * 64;
    unsigned off = OOB;
    if (p < a.npix) {
      const int hx = p % a.HW;
      const int r = p / a.HW;
      const int hy = r % a.HH;
      const int b = r / a.HH;
      const int n = n_base + b;
      const int iy = oy0 * a.stride - a.pad + hy;
      const int ix = ox0 * a.stride - a.pad + hx;
      if ((n < a.N) && (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W))
        off = (unsigned)(((n * a.H + iy) * a.W + ix) * a.cs_in + q * 4) * 4u;
    }
    aoff[it] = off;
  }
  // weights: element e = tid + it*256 -> (tap-quad tq = e / TN, column j = e % TN)
  // -> byte offset (tq*CoutP + n0 + j)*16 inside the stage slab, OOB (zeros) for
  // columns >= CoutP; recomputed per load (TN is a compile-time constant)
#define EGN_BVOFF(IT)                                                   \
  ((n0 + ((tid + (IT)*NTHREADS) % TN)) < a.CoutP                        \
       ? (unsigned)((((tid + (IT)*NTHREADS) / TN) * a.CoutP) + n0 + ((tid + (IT)*NTHREADS) % TN)) * 16u \
       : OOB)

  // A-fragment base pixel (tap 0,0) of this lane for each 16-row sub-tile
  int pixbase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 16 + li;
    int b = m / tile_px;
    const int rem = m - b * tile_px;
    const int y = rem / a.TW;
    const int x = rem - y * a.TW;
    if (b >= a.TNB) b = 0;  // rows beyond the tile: read anything valid, never stored
    pixbase[mt] = (b * a.HH + y * a.stride) * a.HW + x * a.stride;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nspc = (a.taps + a.tps - 1) / a.tps;  // stages per chunk
  const int nstages = a.nchunk * nspc;

  f32x4 ra[A_IT], rb[B_IT];

// issue the loads of stage S into registers (no wait).  soffset (SGPR) = chunk
// channel offset for x, stage slab offset for w.  Lanes whose channel quad lies
// beyond cs_in (last chunk of a cs_in % 16 != 0 tensor) read zeros.
#define EGN_ISSUE(S)                                                                              \
  {                                                                                               \
    const int c_ = (S) / nspc;                                                                    \
    const int g_ = (S) - c_ * nspc;                                                               \
    if (g_ == 0) {                                                                                \
      const bool cpad_ = (c_ * EGN_CK + q * 4) >= a.cs_in;                                        \
      _Pragma("unroll") for (int it = 0; it < A_IT; ++it) ra[it] = __builtin_bit_cast(            \
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, cpad_ ? OOB : aoff[it], c_ * EGN_CK * 4, 0)); \
    }                                                                                             \
    const int sw_ = (c_ * a.taps + g_ * a.tps) * CKQ * a.CoutP * 16;                              \
    _Pragma("unroll") for (int it = 0; it < B_IT; ++it) rb[it] =                                  \
        __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, EGN_BVOFF(it), sw_, 0)); \
  }

  EGN_ISSUE(0)
  for (int s = 0; s < nstages; ++s) {
    __syncthreads();  // every wave finished reading the previous stage
    {  // write the staged registers of stage s to LDS
      const int c = s / nspc;
      const int g = s - c * nspc;
      const int nts = min(a.tps, a.taps - g * a.tps);
      if (g == 0) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
          const int p = p0 + it * 64;
          if (p < a.npix) *reinterpret_cast<f32x4*>(&sA[q * a.npixp + (p ^ (q << 1))]) = ra[it];
        }
      }
      const int b_elems = nts * CKQ * TN;
#pragma unroll
      for (int it = 0; it < B_IT; ++it) {
        const int e = tid + it * NTHREADS;
        if (e < b_elems) *reinterpret_cast<f32x4*>(&sB[e]) = rb[it];
      }
    }
    __syncthreads();
    if (s + 1 < nstages) EGN_ISSUE(s + 1)  // in flight during the MFMA loop below

    const int g = s % nspc;
    const int t0 = g * a.tps;
    const int nts = min(a.tps, a.taps - t0);
    for (int tt = 0; tt < nts; ++tt) {
      const int t = t0 + tt;
      const int ky = t / a.KW;
      const int kx = t - ky * a.KW;
      const int dpix = ky * a.HW + kx;
      float4 af[MT], bf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = sA[kq * a.npixp + ((pixbase[mt] + dpix) ^ (kq << 1))];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = sB[(tt * CKQ + kq) * TN + (wn * NT + nt) * 16 + li];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt].x, bf[nt].x, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt].y, bf[nt].y, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt].z, bf[nt].z, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt].w, bf[nt].w, acc[mt][nt], 0, 0, 0);
        }
    }
  }

  const int act = a.act & EGN_ACT_MASK;
  const bool res_after = (a.act & EGN_ACT_RES_AFTER) != 0;
  const int howo = a.Ho * a.Wo;

  if (!a.out_nchw) {
    // ---- NHWC epilogue through LDS: float4 stores along the channel axis ----
    constexpr int SC_LD = TNW + 4;  // floats per sC row (keeps 16-B alignment, spreads banks)
    __syncthreads();                // main-loop LDS reads are done
    float* sC = reinterpret_cast<float*>(smem) + (size_t)wave * (MT * 16) * SC_LD;
    int* sPix = reinterpret_cast<int*>(reinterpret_cast<float*>(smem) + (size_t)4 * (MT * 16) * SC_LD);
    if (tid < TM) {  // output pixel index of tile row m = tid, -1 = outside
      const int m = tid;
      const int b = m / tile_px;
      const int rem = m - b * tile_px;
      const int y = rem / a.TW;
      const int x = rem - y * a.TW;
      const int n = n_base + b;
      const int oy = oy0 + y;
      const int ox = ox0 + x;
      sPix[m] = (b < a.TNB && n < a.N && oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + (wn * NT + nt) * 16 + li;
      const bool cok = co < a.CoutP;
      const float sc = cok ? a.scale[co] : 0.f;
      const float sh = cok ? a.shift[co] : 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(mt * 16 + kq * 4 + r) * SC_LD + nt * 16 + li] = acc[mt][nt][r] * sc + sh;
    }
    __syncthreads();
    constexpr int C4 = TNW / 4;            // float4 per row of the wave's slab
    constexpr int NV = MT * 16 * C4;       // float4 per wave
    const int cbase = n0 + wn * TNW;
    for (int idx = lane; idx < NV; idx += 64) {
      const int row = idx / C4;
      const int c4 = idx - row * C4;
      const int pix = sPix[wm * MT * 16 + row];
      const int co = cbase + c4 * 4;
      if (pix < 0 || co >= a.cs_out) continue;
      float4 v = *reinterpret_cast<const float4*>(&sC[row * SC_LD + c4 * 4]);
      const size_t gidx = (size_t)pix * a.cs_out + co;
      float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.res) rv = *reinterpret_cast<const float4*>(a.res + gidx);
      if (a.res && !res_after) { v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
      v.x = egn_act(v.x, act); v.y = egn_act(v.y, act); v.z = egn_act(v.z, act); v.w = egn_act(v.w, act);
      if (a.res && res_after) { v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w; }
      // keep pad channels zero
      if (co + 0 >= a.Cout) v.x = 0.f;
      if (co + 1 >= a.Cout) v.y = 0.f;
      if (co + 2 >= a.Cout) v.z = 0.f;
      if (co + 3 >= a.Cout) v.w = 0.f;
      *reinterpret_cast<float4*>(a.y + gidx) = v;
    }
    return;
  }

  // ---- NCHW epilogue (heads, final Linear): lane owns rows 4*kq + r and column li
  const bool tw4 = (a.TW & 3) == 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m0 = (wm * MT + mt) * 16 + kq * 4;
    int on[4], sp[4];  // image index and oy*Wo+ox of each row, sp < 0 = not stored
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r == 0 || !tw4) {
        const int m = m0 + r;
        const int b = m / tile_px;
        const int rem = m - b * tile_px;
        const int y = rem / a.TW;
        const int x = rem - y * a.TW;
        const int oy = oy0 + y;
        const int ox = ox0 + x;
        on[r] = n_base + b;
        sp[r] = (b < a.TNB && on[r] < a.N && oy < a.Ho && ox < a.Wo) ? oy * a.Wo + ox : -1;
        if (tw4) {  // rows 1..3 follow in x
#pragma unroll
          for (int k = 1; k < 4; ++k) {
            on[k] = on[0];
            sp[k] = (sp[0] >= 0 && ox + k < a.Wo) ? sp[0] + k : -1;
          }
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + (wn * NT + nt) * 16 + li;
      if (co >= a.Cout) continue;
      const float sc = a.scale[co];
      const float sh = a.shift[co];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (sp[r] < 0) continue;
        const size_t idx = ((size_t)on[r] * a.Cout + co) * howo + sp[r];
        a.y[idx] = egn_act(acc[mt][nt][r] * sc + sh, act);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// configurations (all 4 waves = 256 threads)
// ---------------------------------------------------------------------------
static const ConvConfig kConfigs[] = {
    // id wm wn mt nt ai bi   (ai / bi = staging depth, must match egn_conv_launch)
    {1, 4, 1, 4, 3, 6, 7},   // 256 x 48   (C = 48 layers)
    {2, 2, 2, 4, 3, 6, 8},   // 128 x 96   (C = 96)
    {3, 2, 2, 4, 2, 8, 8},   // 128 x 64   (C = 64, 192, 256, 384)
    {4, 4, 1, 4, 1, 8, 8},   // 256 x 16
    {5, 4, 1, 4, 2, 8, 8},   // 256 x 32
    {6, 4, 1, 2, 3, 8, 8},   // 128 x 48
    {7, 2, 2, 2, 3, 8, 8},   //  64 x 96
    {8, 2, 2, 2, 2, 8, 8},   //  64 x 64
    {9, 1, 4, 4, 1, 8, 8},   //  64 x 64 (one M strip, N across waves)
    {10, 1, 4, 2, 3, 8, 8},  //  32 x 192
};
static const int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

extern "C" int egn_conv_num_configs(void) { return kNumConfigs; }
const ConvConfig* egn_conv_config(int cfg) {
  return (cfg >= 1 && cfg <= kNumConfigs) ? &kConfigs[cfg - 1] : nullptr;
}
extern "C" int egn_conv_config_info(int cfg, int* tile_m, int* tile_n) {
  if (cfg < 1 || cfg > kNumConfigs) return EGN_E_BADARG;
  if (tile_m) *tile_m = kConfigs[cfg - 1].tile_m();
  if (tile_n) *tile_n = kConfigs[cfg - 1].tile_n();
  return 0;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

static size_t lds_bytes_for(const ConvArgs& a, const ConvConfig& cf) {
  const size_t main_loop = (size_t)(EGN_CKQ * a.npixp + a.tps * EGN_CKQ * cf.tile_n()) * 16;
  // epilogue: 4 waves x (MT*16 rows) x (NT*16 + 4) floats + TM pixel indices
  const size_t epi = a.out_nchw ? 0 : (size_t)4 * cf.mt * 16 * (cf.nt * 16 + 4) * 4 + (size_t)cf.tile_m() * 4;
  return main_loop > epi ? main_loop : epi;
}

// Choose the spatial tile for a config: minimise (MFMA work incl. padding +
// LDS fill work) over power-of-two tile shapes, subject to the LDS budget and
// to the per-lane staging registers (A_IT / B_IT dwordx4 loads per stage).
static bool plan_tile(ConvArgs& a, const ConvConfig& cf, size_t lds_budget, double* cost_out) {
  const int tm = cf.tile_m();
  const int tn = cf.tile_n();
  double best = -1.0;
  ConvArgs bestA = a;
  for (int tw = 1; tw <= 64 && tw <= tm; tw *= 2) {
    if (tw < 4 && tw < a.Wo) continue;  // narrow tiles only for maps that narrow
    for (int th = 1; th * tw <= tm; th *= 2) {
      const int tnb = tm / (tw * th);
      if (tnb * tw * th != tm) continue;
      // no point in tiles much larger than the map
      if (tw >= 2 * a.Wo && tw > 1) continue;
      if (th >= 2 * a.Ho && th > 1) continue;
      ConvArgs c = a;
      c.TH = th; c.TW = tw; c.TNB = tnb;
      c.HH = (th - 1) * a.stride + a.KH;
      c.HW = (tw - 1) * a.stride + a.KW;
      c.npix = tnb * c.HH * c.HW;
      c.npixp = (c.npix + 15) & ~15;
      if (c.npix * EGN_CKQ > cf.ai * 256) continue;
      c.tiles_x = cdiv(a.Wo, tw);
      c.tiles_y = cdiv(a.Ho, th);
      const int tiles_b = cdiv(a.N, tnb);
      // taps per stage: as many as fit the LDS budget and the staging registers
      int tps = a.taps;
      c.tps = tps;
      while (tps > 1 && (lds_bytes_for(c, cf) > lds_budget || tps * EGN_CKQ * tn > cf.bi * 256)) {
        --tps;
        c.tps = tps;
      }
      if (lds_bytes_for(c, cf) > lds_budget || tps * EGN_CKQ * tn > cf.bi * 256) continue;
      // balance the stages (e.g. 9 taps -> 5+4 instead of 8+1)
      const int nst = cdiv(a.taps, tps);
      c.tps = cdiv(a.taps, nst);
      const double tiles = (double)c.tiles_x * c.tiles_y * tiles_b * cdiv(a.CoutP, tn);
      const double mfma = (double)tm * tn * a.taps * EGN_CK;  // per chunk per tile
      const double fill = (double)c.npix * EGN_CK * 24.0 + (double)a.taps * EGN_CK * tn * 12.0;
      // ties (1x1 convs have no halo): prefer contiguous pixels over many images
      const double cost = tiles * (mfma + fill + 4000.0 * nst + 64.0 * tnb + 8.0 * th);
      if (best < 0 || cost < best) { best = cost; bestA = c; }
    }
  }
  if (best < 0) return false;
  a = bestA;
  if (cost_out) *cost_out = best;
  return true;
}

int egn_conv_plan(ConvArgs& a, int& cfg_id, size_t& lds_bytes) {
  if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.Cin <= 0 || a.Cout <= 0) return EGN_E_BADARG;
  if (a.cs_in % 4 || a.cs_in < a.Cin) return EGN_E_BADARG;
  if (!a.out_nchw && (a.cs_out % 4 || a.cs_out < a.Cout)) return EGN_E_BADARG;
  if (a.KH < 1 || a.KW < 1 || a.stride < 1 || a.pad < 0) return EGN_E_BADARG;
  a.Ho = (a.H + 2 * a.pad - a.KH) / a.stride + 1;
  a.Wo = (a.W + 2 * a.pad - a.KW) / a.stride + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return EGN_E_BADARG;
  // 32-bit byte offsets into x (buffer loads) and 32-bit pixel indices
  if ((double)a.N * a.H * a.W * a.cs_in * 4.0 >= 2147483648.0) return EGN_E_BADARG;
  if ((double)a.N * a.Ho * a.Wo >= 2147483648.0) return EGN_E_BADARG;
  a.CoutP = (a.Cout + 15) & ~15;
  a.nchunk = cdiv(a.Cin, EGN_CK);
  a.taps = a.KH * a.KW;
  const size_t budget = 64 * 1024;
  if (cfg_id >= 1 && cfg_id <= kNumConfigs) {
    if (!plan_tile(a, kConfigs[cfg_id - 1], budget, nullptr)) return EGN_E_LDS;
  } else {
    double best = -1.0;
    int best_id = 0;
    ConvArgs bestA = a;
    for (int k = 0; k < kNumConfigs; ++k) {
      ConvArgs c = a;
      double cost;
      if (!plan_tile(c, kConfigs[k], budget, &cost)) continue;
      // mild preference for filling the chip: penalise grids below 256 blocks
      const double blocks = (double)c.tiles_x * c.tiles_y * cdiv(a.N, c.TNB) * cdiv(a.CoutP, kConfigs[k].tile_n());
      if (blocks < 256.0) cost *= 256.0 / blocks > 4.0 ? 4.0 : 256.0 / blocks;
      if (best < 0 || cost < best) { best = cost; best_id = kConfigs[k].id; bestA = c; }
    }
    if (best < 0) return EGN_E_LDS;
    a = bestA;
    cfg_id = best_id;
  }
  lds_bytes = lds_bytes_for(a, kConfigs[cfg_id - 1]);
  return 0;
}

template <int WM, int WN, int MT, int NT, int AI, int BI>
static int launch_one(const ConvArgs& a, size_t lds, hipStream_t stream) {
  const int tiles_b = cdiv(a.N, a.TNB);
  dim3 grid(a.tiles_x * a.tiles_y * tiles_b, cdiv(a.CoutP, WN * NT * 16));
  hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, MT, NT, AI, BI>), grid, dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}

int egn_conv_launch(const ConvArgs& a, int cfg_id, hipStream_t stream) {
  const size_t lds = lds_bytes_for(a, kConfigs[cfg_id - 1]);
  switch (cfg_id) {
    case 1: return launch_one<4, 1, 4, 3, 6, 7>(a, lds, stream);
    case 2: return launch_one<2, 2, 4, 3, 6, 8>(a, lds, stream);
    case 3: return launch_one<2, 2, 4, 2, 8, 8>(a, lds, stream);
    case 4: return launch_one<4, 1, 4, 1, 8, 8>(a, lds, stream);
    case 5: return launch_one<4, 1, 4, 2, 8, 8>(a, lds, stream);
    case 6: return launch_one<4, 1, 2, 3, 8, 8>(a, lds, stream);
    case 7: return launch_one<2, 2, 2, 3, 8, 8>(a, lds, stream);
    case 8: return launch_one<2, 2, 2, 2, 8, 8>(a, lds, stream);
    case 9: return launch_one<1, 4, 4, 1, 8, 8>(a, lds, stream);
    case 10: return launch_one<1, 4, 2, 3, 8, 8>(a, lds, stream);
    default: return EGN_E_BADARG;
  }
}
