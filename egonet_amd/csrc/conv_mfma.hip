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
#include "conv_common.h"

// A_IT / B_IT (template): max dwordx4 loads per lane for the halo tile of a chunk / the weights of a stage

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

  // tile row -> output pixel table for the epilogue (ordered by the K loop's first barrier)
  if (!a.out_nchw) conv_epi_pixels<WM, MT>(a, smem, tid, n_base, oy0, ox0);

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

  if (a.out_nchw) {
    conv_epi_nchw<WM, WN, MT, NT>(a, acc, tid, n_base, oy0, ox0, n0);
  } else {
    // the staging registers are dead here: all residual loads go out together
    ConvEpiRegs<MT, NT> er;
    conv_epi_prefetch<WM, WN, MT, NT>(a, smem, tid, n0, er);
    conv_epi_finish<WM, WN, MT, NT>(a, acc, smem, tid, n0, er);
  }
}

template <int WM, int WN, int MT, int NT, int AI, int BI>
static int launch_one(const ConvArgs& a, size_t lds, hipStream_t stream) {
  const int tiles_b = (a.N + a.TNB - 1) / a.TNB;
  dim3 grid(a.tiles_x * a.tiles_y * tiles_b, (a.CoutP + WN * NT * 16 - 1) / (WN * NT * 16));
  hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, MT, NT, AI, BI>), grid, dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}

// staged family: config ids 1..10 (table in conv_plan.hip)
int egn_conv_launch_staged(const ConvArgs& a, int cfg_id, size_t lds, hipStream_t stream) {
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
