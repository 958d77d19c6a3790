// conv_mfma.hip -- fused convolution for gfx950 as an fp32-MFMA implicit GEMM.
//
//   y = act( conv(x, w) * scale + shift (+ res) )
//
// GEMM view: M = output pixels of a (TNB x TH x TW) tile, N = output channels,
// K = (tap, input channel).  K is walked in chunks of 16 input channels; for a
// chunk the input halo tile of the block is staged ONCE in LDS and re-used by
// every filter tap (the 3x3 stencil re-use happens in LDS, not in HBM), the
// weights of the chunk are staged per group of taps.
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain).  Lane l
// supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it owns
// C[4*(l>>4) + r][l&15], r = 0..3.  Each lane-quad kq = l>>4 is given 4
// consecutive input channels of the chunk, so one ds_read_b128 feeds 4 MFMA
// k-steps (the K order inside a chunk is a permutation of the reference's
// summation order, which fp32 parity allows).
//
// LDS layout (float4 granules, 16 B):
//   sA[q][pos]   q = channel quad 0..3, pos = halo pixel ^ (q<<1)   (XOR swizzle
//                keeps both the ds_write_b128 fill and the ds_read_b128 fragment
//                reads on distinct 16-B slots), plane stride npixp (mult. of 16)
//   sB[t][q][co] weights of tap t, channel quad q, output channel co (TN wide)
//   sOff[p]      element offset of halo pixel p in x, -1 = zero padding
#include "egn_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float egn_act(float v, int act) {
  switch (act) {
    case EGN_ACT_RELU: return fmaxf(v, 0.0f);
    case EGN_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case EGN_ACT_LEAKY: return v > 0.0f ? v : 0.01f * v;
    default: return v;
  }
}

template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(ConvArgs a) {
  constexpr int NTHREADS = 64 * WM * WN;
  constexpr int TN = WN * NT * 16;
  constexpr int CKQ = EGN_CKQ;

  extern __shared__ float4 smem[];
  float4* sA = smem;
  float4* sB = smem + CKQ * a.npixp;
  int* sOff = reinterpret_cast<int*>(sB + a.tps * CKQ * TN);

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

  // halo pixel -> element offset table
  for (int p = tid; p < a.npix; p += NTHREADS) {
    const int hx = p % a.HW;
    const int r = p / a.HW;
    const int hy = r % a.HH;
    const int b = r / a.HH;
    const int n = n_base + b;
    const int iy = oy0 * a.stride - a.pad + hy;
    const int ix = ox0 * a.stride - a.pad + hx;
    const bool ok = (n < a.N) && (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
    sOff[p] = ok ? ((n * a.H + iy) * a.W + ix) * a.cs_in : -1;
  }

  // A-fragment base pixel (tap 0,0) of this lane for each 16-row sub-tile
  int pixbase[MT];
  const int tile_px = a.TH * a.TW;
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

  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(a.w);
  const int a_elems = a.npix * CKQ;

  for (int c = 0; c < a.nchunk; ++c) {
    for (int t0 = 0; t0 < a.taps; t0 += a.tps) {
      __syncthreads();  // previous stage fully consumed (also orders sOff)
      if (t0 == 0) {
        const int cbase = c * EGN_CK;
        for (int e = tid; e < a_elems; e += NTHREADS) {
          const int q = e & (CKQ - 1);
          const int p = e >> 2;
          const int off = sOff[p];
          const int ci = cbase + q * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (off >= 0 && ci < a.cs_in) v = *reinterpret_cast<const float4*>(a.x + off + ci);
          sA[q * a.npixp + (p ^ (q << 1))] = v;
        }
      }
      const int nts = min(a.tps, a.taps - t0);
      const int b_elems = nts * CKQ * TN;
      const size_t wbase = (size_t)(c * a.taps + t0) * CKQ * a.CoutP;
      for (int e = tid; e < b_elems; e += NTHREADS) {
        const int j = e % TN;
        const int tq = e / TN;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + j < a.CoutP) v = w4[wbase + (size_t)tq * a.CoutP + n0 + j];
        sB[e] = v;
      }
      __syncthreads();

      for (int tt = 0; tt < nts; ++tt) {
        const int t = t0 + tt;
        const int ky = t / a.KW;
        const int kx = t - ky * a.KW;
        const int dpix = ky * a.HW + kx;
        float4 af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          af[mt] = sA[kq * a.npixp + ((pixbase[mt] + dpix) ^ (kq << 1))];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bf[nt] = sB[(tt * CKQ + kq) * TN + (wn * NT + nt) * 16 + li];
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
  }

  const int act = a.act & EGN_ACT_MASK;
  const bool res_after = (a.act & EGN_ACT_RES_AFTER) != 0;
  // epilogue: lane owns rows 4*kq + r (r = 0..3) and column li of every 16x16
  // sub-tile.  With TW % 4 == 0 the 4 rows are consecutive pixels of one output
  // row (one decomposition); otherwise each row is decomposed on its own.
  const int howo = a.Ho * a.Wo;
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
          for (int q = 1; q < 4; ++q) {
            on[q] = on[0];
            sp[q] = (sp[0] >= 0 && ox + q < a.Wo) ? sp[0] + q : -1;
          }
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + (wn * NT + nt) * 16 + li;
      if (co >= a.CoutP) continue;
      const float sc = a.scale[co];
      const float sh = a.shift[co];
      if (a.out_nchw) {
        if (co >= a.Cout) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (sp[r] < 0) continue;
          const size_t idx = ((size_t)on[r] * a.Cout + co) * howo + sp[r];
          a.y[idx] = egn_act(acc[mt][nt][r] * sc + sh, act);
        }
      } else {
        if (co >= a.cs_out) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (sp[r] < 0) continue;
          const size_t idx = ((size_t)on[r] * howo + sp[r]) * a.cs_out + co;
          float v = acc[mt][nt][r] * sc + sh;
          if (a.res && !res_after) v += a.res[idx];
          v = egn_act(v, act);
          if (a.res && res_after) v = a.res[idx] + v;
          if (co >= a.Cout) v = 0.0f;  // keep pad channels zero
          a.y[idx] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// configurations
// ---------------------------------------------------------------------------
static const ConvConfig kConfigs[] = {
    {1, 4, 1, 4, 3},  // 256 x 48   (C = 48 layers)
    {2, 2, 2, 4, 3},  // 128 x 96   (C = 96)
    {3, 2, 2, 4, 2},  // 128 x 64   (C = 64, 192, 256, 384)
    {4, 4, 1, 4, 1},  // 256 x 16
    {5, 4, 1, 4, 2},  // 256 x 32
    {6, 4, 1, 2, 3},  // 128 x 48
    {7, 2, 2, 2, 3},  //  64 x 96
    {8, 2, 2, 2, 2},  //  64 x 64
    {9, 1, 4, 4, 1},  //  64 x 64 (one M strip, N across waves)
    {10, 1, 4, 2, 3}, //  32 x 192
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

static size_t lds_bytes_for(const ConvArgs& a, int tn) {
  return (size_t)(EGN_CKQ * a.npixp + a.tps * EGN_CKQ * tn) * 16 + (size_t)a.npix * 4;
}

// Choose the spatial tile for a config: minimise (MFMA work incl. padding +
// LDS fill work) over power-of-two tile shapes with TW % 4 == 0.
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
      c.tiles_x = cdiv(a.Wo, tw);
      c.tiles_y = cdiv(a.Ho, th);
      const int tiles_b = cdiv(a.N, tnb);
      // taps per stage: as many as fit the budget
      int tps = a.taps;
      c.tps = tps;
      while (tps > 1 && lds_bytes_for(c, tn) > lds_budget) { --tps; c.tps = tps; }
      if (lds_bytes_for(c, tn) > lds_budget) continue;
      // balance the stages (e.g. 9 taps -> 3x3 instead of 8+1)
      const int nst = cdiv(a.taps, tps);
      c.tps = cdiv(a.taps, nst);
      const double tiles = (double)c.tiles_x * c.tiles_y * tiles_b * cdiv(a.CoutP, tn);
      const double mfma = (double)tm * tn * a.taps * EGN_CK;           // per chunk per tile
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
  if ((double)a.N * a.H * a.W * a.cs_in >= 2147483648.0) return EGN_E_BADARG;
  a.CoutP = (a.Cout + 15) & ~15;
  a.nchunk = cdiv(a.Cin, EGN_CK);
  a.taps = a.KH * a.KW;
  const size_t budget = 64 * 1024;
  if (cfg_id >= 1 && cfg_id <= kNumConfigs) {
    if (!plan_tile(a, kConfigs[cfg_id - 1], 160 * 1024 - 256, nullptr)) return EGN_E_LDS;
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
  lds_bytes = lds_bytes_for(a, kConfigs[cfg_id - 1].tile_n());
  return 0;
}

template <int WM, int WN, int MT, int NT>
static int launch_one(const ConvArgs& a, size_t lds, hipStream_t stream) {
  auto kern = conv_mfma_kernel<WM, WN, MT, NT>;
  if (lds > 64 * 1024) {
    // only explicit configs can exceed the default dynamic-LDS limit
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  }
  const int tiles_b = cdiv(a.N, a.TNB);
  dim3 grid(a.tiles_x * a.tiles_y * tiles_b, cdiv(a.CoutP, WN * NT * 16));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, stream, a);
  return (int)hipGetLastError();
}

int egn_conv_launch(const ConvArgs& a, int cfg_id, hipStream_t stream) {
  const size_t lds = lds_bytes_for(a, kConfigs[cfg_id - 1].tile_n());
  switch (cfg_id) {
    case 1: return launch_one<4, 1, 4, 3>(a, lds, stream);
    case 2: return launch_one<2, 2, 4, 3>(a, lds, stream);
    case 3: return launch_one<2, 2, 4, 2>(a, lds, stream);
    case 4: return launch_one<4, 1, 4, 1>(a, lds, stream);
    case 5: return launch_one<4, 1, 4, 2>(a, lds, stream);
    case 6: return launch_one<4, 1, 2, 3>(a, lds, stream);
    case 7: return launch_one<2, 2, 2, 3>(a, lds, stream);
    case 8: return launch_one<2, 2, 2, 2>(a, lds, stream);
    case 9: return launch_one<1, 4, 4, 1>(a, lds, stream);
    case 10: return launch_one<1, 4, 2, 3>(a, lds, stream);
    default: return EGN_E_BADARG;
  }
}
