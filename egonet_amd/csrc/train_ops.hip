// train_ops.hip -- HBM-bound building blocks of the training step (reference:
// libs/trainer/trainer.py:183-209 zero_grad / forward / loss / backward / Adam
// step; libs/model/FCmodel.py Linear + BatchNorm1d + ReLU + Dropout;
// libs/loss/function.py:204-215 MSELoss1D; libs/optimizer/optimizer.py:8-40).
//
// The GEMMs of forward, dgrad and wgrad all run on the fp32-MFMA conv kernel
// (a Linear is a 1x1 conv):  z = x W^T     -> weights packed from W
//                            dx = dz W     -> weights packed from W^T
//                            dW = dz^T x   -> "pixels" = rows of dz^T, weights packed from x^T
// so what lives here is: on-device weight packing (every step, the weights
// change), LDS-tiled transposes, per-column batch statistics, the fused
// BatchNorm(+ReLU+dropout) forward / backward, column sums, MSE and Adam.
// Activations are row-major [rows, ld] fp32 = NHWC [rows,1,1,ld]; ld % 4 == 0.
#include "egn_internal.h"
#include "wino4_pack.h"

static inline int grid_for(size_t work_items, int block) {
  size_t g = (work_items + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------
// pack a row-major matrix as conv weights [nchunk][1][4][CoutP][4]
//   transpose == 0: W[co][ci] = src[co*ld + ci]   (rows = Cout)
//   transpose == 1: W[co][ci] = src[ci*ld + co]   (rows = Cin)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_matrix_kernel(const float* __restrict__ src, int ld, int cout, int cin,
                                                          int transpose, float4* __restrict__ dst, int coutp,
                                                          int nchunk) {
  const size_t total = (size_t)nchunk * 4 * coutp;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(e % coutp);
    const int cq = (int)(e / coutp);  // chunk*4 + quad
    const int ci0 = cq * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (co < cout) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + r;
        if (ci < cin) v[r] = transpose ? src[(size_t)ci * ld + co] : src[(size_t)co * ld + ci];
      }
    }
    dst[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" int egn_pack_matrix_f32(const float* src, int ld, int cout, int cin, int transpose, float* dst,
                                   void* stream) {
  if (cout <= 0 || cin <= 0 || ld < (transpose ? cout : cin)) return EGN_E_BADARG;
  const int coutp = (cout + 15) & ~15;
  const int nchunk = (cin + EGN_CK - 1) / EGN_CK;
  const size_t total = (size_t)nchunk * 4 * coutp;
  hipLaunchKernelGGL(pack_matrix_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, ld,
                     cout, cin, transpose, reinterpret_cast<float4*>(dst), coutp, nchunk);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// dst[c][r] = src[r][c]  (src [R, lds], dst [C, ldd]); columns r >= R of dst up
// to ldd are zeroed.  32x32 LDS tiles (+1 pad), coalesced on both sides.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int R, int C, int lds_,
                                                        float* __restrict__ dst, int ldd) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int r = r0 + ty + k, c = c0 + tx;
    tile[ty + k][tx] = (r < R && c < C) ? src[(size_t)r * lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int c = c0 + ty + k, r = r0 + tx;
    if (c < C && r < ldd) dst[(size_t)c * ldd + r] = tile[tx][ty + k];
  }
}

extern "C" int egn_transpose_f32(const float* src, int R, int C, int ld_src, float* dst, int ld_dst, void* stream) {
  if (R <= 0 || C <= 0 || ld_src < C || ld_dst < R) return EGN_E_BADARG;
  dim3 grid((C + 31) / 32, (ld_dst + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, R, C, ld_src, dst, ld_dst);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// column reductions over the rows of a row-major [rows, ld] matrix (ld % 4 == 0;
// NHWC activations are exactly this with rows = N*H*W, ld = cs).  A block covers
// up to 64 float4 column groups and as many row lanes as fit in 256 threads, so
// consecutive lanes read consecutive 16-byte pieces of consecutive rows; the
// rows are split over gridDim.y blocks whose partials (double) are combined in
// a fixed order by a second kernel (deterministic).
//   mode 0: sum[c]              = sum_r a[r][c]
//   mode 1: stats of z          : mean[c], invstd[c] (biased var + eps), var_unbiased[c],
//                                 optional running-stat update (torch momentum rule)
//   mode 2: BN backward sums    : s1[c] = sum dpre, s2[c] = sum dpre * xhat
//           dpre = dy * (mask? mask*keep_scale : 1) * gate,  gate = (gamma*xhat + beta + res > 0) or 1
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Dropout keep masks drawn IN the kernels (FCmodel.py:24, 38-41: nn.Dropout(p) after every ReLU of the lifter):
// Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, counter = (float4 group index, layer, *step).  The
// forward kernel and the two backward kernels of a unit regenerate the same mask from the same arguments, so no
// mask tensor exists and no RNG kernel runs inside the step; `step` is read from device memory (hipGraph-safe:
// the optimizer's step counter).  element kept  <=>  its 32-bit draw >= p * 2^32.
// ---------------------------------------------------------------------------
struct DropArgs {
  unsigned thresh;           // p * 2^32 (0 = no RNG dropout)
  unsigned seed_lo, seed_hi;
  unsigned layer;
  const int* step;
};

__device__ __forceinline__ void egn_philox4(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                            unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// keep mask (0 / 1) of float4 group e4
__device__ __forceinline__ void egn_drop_mask4(const DropArgs& d, unsigned step, size_t e4, float (&m)[4]) {
  unsigned r[4];
  egn_philox4(d.seed_lo, d.seed_hi, (unsigned)e4, (unsigned)(e4 >> 32), d.layer, step, r);
#pragma unroll
  for (int k = 0; k < 4; ++k) m[k] = r[k] >= d.thresh ? 1.f : 0.f;
}

struct ColArgs {
  DropArgs drop;       // mode 2: keep mask drawn in the kernel when mask == NULL and drop.thresh != 0
  const float* a;      // z (mode 1, 2) or the matrix to sum (mode 0)
  const float* dy;     // mode 2
  const float* mask;   // mode 2, optional dropout keep mask (0/1)
  const float* mean;   // mode 2
  const float* invstd; // mode 2
  const float* gamma;  // mode 2
  const float* beta;   // mode 2
  const float* res;    // mode 2, optional residual added before the ReLU gate
  float* run_mean;     // mode 1, optional running statistics to update in place
  float* run_var;
  float momentum;
  float* out0;         // sum | mean | s1
  float* out1;         // - | invstd | s2
  float* out2;         // - | unbiased var | -
  int rows, cols, ld;
  int mode, relu;
  float eps, keep_scale;
  int nsplit;
};

constexpr int EGN_COL_MAX_SPLITS = 256;

__global__ __launch_bounds__(256) void colreduce_partial_kernel(ColArgs p, double* __restrict__ ws) {
  __shared__ double red[2][256][4];
  const int cg_total = p.ld / 4;
  const int g0 = blockIdx.x * 64;
  const int cgb = min(64, cg_total - g0);  // float4 column groups of this block
  const int RL = 256 / cgb;                // row lanes
  const int t = threadIdx.x;
  const int rl = t / cgb, g = t - rl * cgb;
  const int rows_per = (p.rows + p.nsplit - 1) / p.nsplit;
  const int r_lo = blockIdx.y * rows_per;
  const int r_hi = min(p.rows, r_lo + rows_per);
  const int c0 = (g0 + g) * 4;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (rl < RL) {
    float mean[4], istd[4], gm[4], bt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = p.mode == 2 && c0 + k < p.cols;
      mean[k] = ok ? p.mean[c0 + k] : 0.f;
      istd[k] = ok ? p.invstd[c0 + k] : 0.f;
      gm[k] = ok ? p.gamma[c0 + k] : 0.f;
      bt[k] = ok ? p.beta[c0 + k] : 0.f;
    }
    float f0[4] = {0, 0, 0, 0}, f1[4] = {0, 0, 0, 0};
    int cnt = 0;
    const bool rng = p.mode == 2 && !p.mask && p.drop.thresh != 0;
    const unsigned dstep = rng ? (unsigned)*p.drop.step : 0u;
    // four rows per round: their loads are issued together (one block per CU and a chain
    // of dependent round trips otherwise -- the kernel is latency bound, not bandwidth bound)
    for (int r = r_lo + rl; r < r_hi; r += 4 * RL) {
      float4 v4[4], d4[4], m4[4], r4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + u * RL;
        const bool ok = rr < r_hi;
        const size_t i = (size_t)(ok ? rr : r) * p.ld + c0;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        v4[u] = ok ? *reinterpret_cast<const float4*>(p.a + i) : zero;
        if (p.mode == 2) {
          d4[u] = ok ? *reinterpret_cast<const float4*>(p.dy + i) : zero;   // dy = 0: the row adds nothing
          if (p.mask) m4[u] = *reinterpret_cast<const float4*>(p.mask + i);
          if (p.res) r4[u] = *reinterpret_cast<const float4*>(p.res + i);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
        if (p.mode == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) f0[k] += v[k];
        } else if (p.mode == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { f0[k] += v[k]; f1[k] += v[k] * v[k]; }
        } else {
          float d[4] = {d4[u].x, d4[u].y, d4[u].z, d4[u].w};
          if (p.mask) {
            d[0] *= m4[u].x * p.keep_scale; d[1] *= m4[u].y * p.keep_scale;
            d[2] *= m4[u].z * p.keep_scale; d[3] *= m4[u].w * p.keep_scale;
          } else if (rng) {
            float mk[4];
            const int rr = r + u * RL;
            egn_drop_mask4(p.drop, dstep, ((size_t)(rr < r_hi ? rr : r) * p.ld + c0) / 4, mk);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] *= mk[k] * p.keep_scale;
          }
          float rs[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.res) { rs[0] = r4[u].x; rs[1] = r4[u].y; rs[2] = r4[u].z; rs[3] = r4[u].w; }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float xhat = (v[k] - mean[k]) * istd[k];
            float dd = d[k];
            if (p.relu && !(gm[k] * xhat + bt[k] + rs[k] > 0.f)) dd = p.relu == 2 ? 0.01f * dd : 0.f;
            f0[k] += dd;
            f1[k] += dd * xhat;
          }
        }
      }
      cnt += 4;
      if (cnt >= 64) {  // flush the fp32 partials to double every 64 terms
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += f0[k]; s1[k] += f1[k]; f0[k] = 0.f; f1[k] = 0.f; }
        cnt = 0;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s0[k] += f0[k]; s1[k] += f1[k]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[0][t][k] = s0[k]; red[1][t][k] = s1[k]; }
  __syncthreads();
  // thread (which, g, k) sums its column over the row lanes in lane order
  if (t < cgb * 4) {
    const int gg = t >> 2, k = t & 3;
    const int c = (g0 + gg) * 4 + k;
    if (c < p.cols) {
      double a0 = 0.0, a1 = 0.0;
      for (int l = 0; l < RL; ++l) { a0 += red[0][l * cgb + gg][k]; a1 += red[1][l * cgb + gg][k]; }
      ws[((size_t)blockIdx.y * 2 + 0) * p.cols + c] = a0;
      ws[((size_t)blockIdx.y * 2 + 1) * p.cols + c] = a1;
    }
  }
}

// stage 2: a block combines the partials of 8 columns with 32 split lanes each
// (fixed association order: lane-strided sums, then the lanes in order) and finalises
__global__ __launch_bounds__(256) void colreduce_final_kernel(ColArgs p, const double* __restrict__ ws) {
  __shared__ double r0[32][9];
  __shared__ double r1[32][9];
  const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double a0 = 0.0, a1 = 0.0;
  if (c < p.cols) {
    // nsplit <= 256: at most 8 rows per lane.  All 16 loads are issued before the first add (the rolled loop made 8
    // dependent L2 round trips: 5.1 us per launch for a kernel the training step runs 610 times, profiles/
    // r3_train_hc_kernel_stats_single_stream.csv); the sums are taken in the same order as before.
    static_assert(EGN_COL_MAX_SPLITS <= 8 * 32, "eight rows per split lane");
    for (int k = sl + 256; k < p.nsplit; k += 32) {      // (tables with more than 256 rows: a large-M GEMM epilogue)
      a0 += ws[((size_t)k * 2 + 0) * p.cols + c];
      a1 += ws[((size_t)k * 2 + 1) * p.cols + c];
    }
    double v0[8], v1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = sl + 32 * u;
      const bool in = k < p.nsplit;
      const size_t kk = in ? (size_t)k : 0;
      v0[u] = ws[(kk * 2 + 0) * p.cols + c];
      v1[u] = ws[(kk * 2 + 1) * p.cols + c];
      if (!in) { v0[u] = 0.0; v1[u] = 0.0; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (sl + 32 * u < p.nsplit) { a0 += v0[u]; a1 += v1[u]; }
    }
  }
  r0[sl][cl] = a0;
  r1[sl][cl] = a1;
  __syncthreads();
  if (sl == 0 && c < p.cols) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < 32; ++k) {
      t0 += r0[k][cl];
      t1 += r1[k][cl];
    }
    if (p.mode == 0) {
      p.out0[c] = (float)t0;
    } else if (p.mode == 1) {
      const double m = t0 / p.rows;
      double var = t1 / p.rows - m * m;
      if (var < 0) var = 0;
      p.out0[c] = (float)m;
      p.out1[c] = (float)(1.0 / sqrt(var + (double)p.eps));
      const float vu = (float)(p.rows > 1 ? var * p.rows / (p.rows - 1) : var);
      if (p.out2) p.out2[c] = vu;
      if (p.run_mean) p.run_mean[c] = (1.f - p.momentum) * p.run_mean[c] + p.momentum * (float)m;
      if (p.run_var) p.run_var[c] = (1.f - p.momentum) * p.run_var[c] + p.momentum * vu;
    } else {
      p.out0[c] = (float)t0;
      p.out1[c] = (float)t1;
    }
  }
}

// ws: caller-provided scratch of egn_colreduce_ws_bytes(cols) bytes
extern "C" long egn_colreduce_ws_bytes(int cols) {
  return (long)EGN_COL_MAX_SPLITS * 2 * cols * (long)sizeof(double);
}

static int launch_col(ColArgs& p, double* ws, void* stream) {
  if (p.rows <= 0 || p.cols <= 0 || p.ld < p.cols || p.ld % 4 || !ws) return EGN_E_BADARG;
  const int cg = p.ld / 4;
  const int gx = (cg + 63) / 64;
  const int rl = 256 / (cg < 64 ? cg : 64);
  int want = (512 + gx - 1) / gx;
  int cap = p.rows / (2 * rl);
  p.nsplit = want < cap ? want : cap;
  if (p.nsplit > EGN_COL_MAX_SPLITS) p.nsplit = EGN_COL_MAX_SPLITS;
  if (p.nsplit < 1) p.nsplit = 1;
  hipLaunchKernelGGL(colreduce_partial_kernel, dim3(gx, p.nsplit), dim3(256), 0, (hipStream_t)stream, p, ws);
  hipLaunchKernelGGL(colreduce_final_kernel, dim3((p.cols + 7) / 8), dim3(256), 0, (hipStream_t)stream, p, ws);
  return (int)hipGetLastError();
}

extern "C" int egn_colsum_f32(const float* a, int rows, int cols, int ld, float* sum, void* ws, void* stream) {
  ColArgs p = {};
  p.a = a; p.out0 = sum; p.rows = rows; p.cols = cols; p.ld = ld; p.mode = 0;
  return launch_col(p, (double*)ws, stream);
}

extern "C" int egn_bn_stats_f32(const float* z, int rows, int cols, int ld, float eps, float* mean, float* invstd,
                                float* var_unbiased, float* running_mean, float* running_var, float momentum,
                                void* ws, void* stream) {
  ColArgs p = {};
  p.a = z; p.out0 = mean; p.out1 = invstd; p.out2 = var_unbiased;
  p.run_mean = running_mean; p.run_var = running_var; p.momentum = momentum;
  p.rows = rows; p.cols = cols; p.ld = ld; p.mode = 1; p.eps = eps;
  return launch_col(p, (double*)ws, stream);
}

// BatchNorm statistics from partial rows [nrows][2][cols] (doubles: sum, sum of squares) that a
// convolution epilogue wrote (egn_conv2d_bnstats_f32): the finalise stage of egn_bn_stats_f32 alone
extern "C" int egn_bn_stats_finalize_f32(const double* partials, long nrows, int rows, int cols, float eps,
                                         float* mean, float* invstd, float* var_unbiased, float* running_mean,
                                         float* running_var, float momentum, void* stream) {
  if (!partials || nrows <= 0 || nrows > 0x7fffffffL || rows <= 0 || cols <= 0 || !mean || !invstd)
    return EGN_E_BADARG;
  ColArgs p = {};
  p.out0 = mean; p.out1 = invstd; p.out2 = var_unbiased;
  p.run_mean = running_mean; p.run_var = running_var; p.momentum = momentum;
  p.rows = rows; p.cols = cols; p.ld = cols; p.mode = 1; p.eps = eps;
  p.nsplit = (int)nrows;
  hipLaunchKernelGGL(colreduce_final_kernel, dim3((cols + 7) / 8), dim3(256), 0, (hipStream_t)stream, p, partials);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// y = act(gamma * (z - mean) * invstd + beta) * (mask ? mask * keep_scale : 1)
// (BatchNorm with the given statistics + ReLU/LeakyReLU + inverted dropout)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ mask, float keep_scale, int relu,
                                                         const float* __restrict__ res, float* __restrict__ y,
                                                         int rows, int cols, int ld, DropArgs drop) {
  // a thread keeps ONE float4 column group for all its rows, so the per-channel
  // parameters are loaded once into registers; a block covers up to 256 column groups
  // and 256 / groups rows per step: consecutive lanes read consecutive 16-byte pieces of
  // consecutive rows (ld == row length for NHWC activations)
  const int ld4 = ld / 4;
  const int cgb = min(256, ld4 - (int)blockIdx.y * 256);
  const int rpi = 256 / cgb;
  const int r_local = threadIdx.x / cgb;
  if (r_local >= rpi) return;
  const int c4 = blockIdx.y * 256 + threadIdx.x - r_local * cgb;
  float pm[4], pi[4], pg[4], pb[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c4 * 4 + k;
    ok[k] = c < cols;
    pm[k] = ok[k] ? mean[c] : 0.f;
    pi[k] = ok[k] ? invstd[c] : 0.f;
    pg[k] = ok[k] ? gamma[c] : 0.f;
    pb[k] = ok[k] ? beta[c] : 0.f;
  }
  const bool rng = !mask && drop.thresh != 0;
  const unsigned dstep = rng ? (unsigned)*drop.step : 0u;
  // `relu`: 0 none, 1 ReLU, 2 LeakyReLU; | 0x10: `res` is added AFTER activation and dropout (the residual block's
  // output, so that the lifter step needs no add pass) instead of in front of the activation (HRNet's BasicBlock)
  const int act = relu & 0xf;
  const bool res_after = (relu & 0x10) != 0;
  for (int r = blockIdx.x * rpi + r_local; r < rows; r += gridDim.x * rpi) {
    const size_t e = (size_t)r * ld4 + c4;
    const float4 v = reinterpret_cast<const float4*>(z)[e];
    const float in[4] = {v.x, v.y, v.z, v.w};
    float out[4];
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask) mk = reinterpret_cast<const float4*>(mask)[e];
    float mka[4] = {mk.x, mk.y, mk.z, mk.w};
    if (rng) egn_drop_mask4(drop, dstep, e, mka);
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res) rv = reinterpret_cast<const float4*>(res)[e];
    const float ra[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o = 0.f;
      if (ok[k]) {
        o = pg[k] * ((in[k] - pm[k]) * pi[k]) + pb[k] + (res_after ? 0.f : ra[k]);
        if (act == 2) o = o > 0.f ? o : 0.01f * o;   // nn.LeakyReLU() default slope (FCmodel.py:19-22)
        else if (act) o = fmaxf(o, 0.f);
        if (mask || rng) o *= mka[k] * keep_scale;
        if (res_after) o = ra[k] + o;                // FCmodel.py:49-51: out = x + dropout(relu(bn(z)))
      }
      out[k] = o;
    }
    reinterpret_cast<float4*>(y)[e] = make_float4(out[0], out[1], out[2], out[3]);
  }
}

// grid of the row-streaming kernels above / below: x = row steps (capped), y = column blocks
static dim3 rowstream_grid(int rows, int ld) {
  const int ld4 = ld / 4;
  const int gy = (ld4 + 255) / 256;
  const int rpi = 256 / (ld4 < 256 ? ld4 : 256);
  int gx = (rows + rpi - 1) / rpi;
  if (gx > 2048) gx = 2048;
  return dim3(gx, gy);
}

static DropArgs make_drop(float p, unsigned long long seed, const int* step, int layer) {
  DropArgs d = {};
  if (p > 0.f && step) {
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
    if (d.thresh == 0) d.thresh = 1;
    d.seed_lo = (unsigned)seed; d.seed_hi = (unsigned)(seed >> 32);
    d.layer = (unsigned)layer; d.step = step;
  }
  return d;
}

extern "C" int egn_bn_act_fwd_f32(const float* z, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, const float* mask, float keep_scale, int relu,
                                  const float* res, float* y, int rows, int cols, int ld, void* stream) {
  if (rows <= 0 || cols <= 0 || ld % 4 || ld < cols) return EGN_E_BADARG;
  hipLaunchKernelGGL(bn_act_fwd_kernel, rowstream_grid(rows, ld), dim3(256), 0, (hipStream_t)stream, z, mean, invstd,
                     gamma, beta, mask, keep_scale, relu, res, y, rows, cols, ld, DropArgs{});
  return (int)hipGetLastError();
}
// ... with the dropout keep mask drawn in the kernel (Philox on (seed; element, layer, *step_dev)); keep_scale = 1/(1-p)
extern "C" int egn_bn_act_fwd_drop_f32(const float* z, const float* mean, const float* invstd, const float* gamma,
                                       const float* beta, float p, unsigned long long seed, const int* step_dev,
                                       int layer, int relu, const float* res, float* y, int rows, int cols, int ld,
                                       void* stream) {
  if (rows <= 0 || cols <= 0 || ld % 4 || ld < cols || !(p >= 0.f && p < 1.f) || (p > 0.f && !step_dev)) return EGN_E_BADARG;
  hipLaunchKernelGGL(bn_act_fwd_kernel, rowstream_grid(rows, ld), dim3(256), 0, (hipStream_t)stream, z, mean, invstd,
                     gamma, beta, (const float*)nullptr, 1.0f / (1.0f - p), relu, res, y, rows, cols, ld,
                     make_drop(p, seed, step_dev, layer));
  return (int)hipGetLastError();
}

extern "C" int egn_bn_bwd_sums_f32(const float* dy, const float* z, const float* mask, float keep_scale,
                                   const float* mean, const float* invstd, const float* gamma, const float* beta,
                                   int relu, const float* res, int rows, int cols, int ld, float* dbeta,
                                   float* dgamma, void* ws, void* stream) {
  ColArgs p = {};
  p.res = res;
  p.a = z; p.dy = dy; p.mask = mask; p.keep_scale = keep_scale; p.mean = mean; p.invstd = invstd;
  p.gamma = gamma; p.beta = beta; p.relu = relu; p.out0 = dbeta; p.out1 = dgamma;
  p.rows = rows; p.cols = cols; p.ld = ld; p.mode = 2;
  return launch_col(p, (double*)ws, stream);
}

extern "C" int egn_bn_bwd_sums_drop_f32(const float* dy, const float* z, float p, unsigned long long seed,
                                        const int* step_dev, int layer, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, int relu, const float* res, int rows,
                                        int cols, int ld, float* dbeta, float* dgamma, void* ws, void* stream) {
  if (!(p >= 0.f && p < 1.f) || (p > 0.f && !step_dev)) return EGN_E_BADARG;
  ColArgs q = {};
  q.drop = make_drop(p, seed, step_dev, layer);
  q.res = res;
  q.a = z; q.dy = dy; q.keep_scale = 1.0f / (1.0f - p); q.mean = mean; q.invstd = invstd;
  q.gamma = gamma; q.beta = beta; q.relu = relu; q.out0 = dbeta; q.out1 = dgamma;
  q.rows = rows; q.cols = cols; q.ld = ld; q.mode = 2;
  return launch_col(q, (double*)ws, stream);
}

// dz = gamma * invstd * (dpre - dbeta/rows - xhat * dgamma/rows)   (batch-stat BN backward)
__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                        const float* __restrict__ mask, float keep_scale,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int relu,
                                                        const float* __restrict__ res,
                                                        const float* __restrict__ dbeta,
                                                        const float* __restrict__ dgamma, float* __restrict__ dz,
                                                        float* __restrict__ dres, int rows, int cols, int ld,
                                                        DropArgs drop) {
  // same row-streaming thread mapping as bn_act_fwd_kernel: per-channel values in registers
  const int ld4 = ld / 4;
  const float inv_rows = 1.0f / (float)rows;
  const int cgb = min(256, ld4 - (int)blockIdx.y * 256);
  const int rpi = 256 / cgb;
  const int r_local = threadIdx.x / cgb;
  if (r_local >= rpi) return;
  const int c4 = blockIdx.y * 256 + threadIdx.x - r_local * cgb;
  float pm[4], pi[4], pg[4], pb[4], pdb[4], pdg[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c4 * 4 + k;
    ok[k] = c < cols;
    pm[k] = ok[k] ? mean[c] : 0.f;
    pi[k] = ok[k] ? invstd[c] : 0.f;
    pg[k] = ok[k] ? gamma[c] : 0.f;
    pb[k] = ok[k] ? beta[c] : 0.f;
    pdb[k] = ok[k] ? dbeta[c] : 0.f;
    pdg[k] = ok[k] ? dgamma[c] : 0.f;
  }
  const bool rng = !mask && drop.thresh != 0;
  const unsigned dstep = rng ? (unsigned)*drop.step : 0u;
  for (int r = blockIdx.x * rpi + r_local; r < rows; r += gridDim.x * rpi) {
    const size_t e = (size_t)r * ld4 + c4;
    const float4 zv = reinterpret_cast<const float4*>(z)[e];
    const float4 dv = reinterpret_cast<const float4*>(dy)[e];
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask) mk = reinterpret_cast<const float4*>(mask)[e];
    const float zi[4] = {zv.x, zv.y, zv.z, zv.w};
    const float di[4] = {dv.x, dv.y, dv.z, dv.w};
    float mi[4] = {mk.x, mk.y, mk.z, mk.w};
    if (rng) egn_drop_mask4(drop, dstep, e, mi);
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res) rv = reinterpret_cast<const float4*>(res)[e];
    const float ri[4] = {rv.x, rv.y, rv.z, rv.w};
    float out[4], gated[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o = 0.f, d = 0.f;
      if (ok[k]) {
        const float xhat = (zi[k] - pm[k]) * pi[k];
        d = di[k];
        if (mask || rng) d *= mi[k] * keep_scale;
        if (relu && !(pg[k] * xhat + pb[k] + ri[k] > 0.f)) d = relu == 2 ? 0.01f * d : 0.f;
        o = pg[k] * pi[k] * (d - pdb[k] * inv_rows - xhat * pdg[k] * inv_rows);
      }
      out[k] = o;
      gated[k] = d;
    }
    reinterpret_cast<float4*>(dz)[e] = make_float4(out[0], out[1], out[2], out[3]);
    if (dres) reinterpret_cast<float4*>(dres)[e] = make_float4(gated[0], gated[1], gated[2], gated[3]);
  }
}

extern "C" int egn_bn_bwd_dz_f32(const float* dy, const float* z, const float* mask, float keep_scale,
                                 const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 int relu, const float* res, const float* dbeta, const float* dgamma, float* dz,
                                 float* dres, int rows, int cols, int ld, void* stream) {
  if (rows <= 0 || cols <= 0 || ld % 4 || ld < cols) return EGN_E_BADARG;
  hipLaunchKernelGGL(bn_bwd_dz_kernel, rowstream_grid(rows, ld), dim3(256), 0, (hipStream_t)stream, dy, z, mask,
                     keep_scale, mean, invstd, gamma, beta, relu, res, dbeta, dgamma, dz, dres, rows, cols, ld,
                     DropArgs{});
  return (int)hipGetLastError();
}
extern "C" int egn_bn_bwd_dz_drop_f32(const float* dy, const float* z, float p, unsigned long long seed,
                                      const int* step_dev, int layer, const float* mean, const float* invstd,
                                      const float* gamma, const float* beta, int relu, const float* res,
                                      const float* dbeta, const float* dgamma, float* dz, float* dres, int rows,
                                      int cols, int ld, void* stream) {
  if (rows <= 0 || cols <= 0 || ld % 4 || ld < cols || !(p >= 0.f && p < 1.f) || (p > 0.f && !step_dev)) return EGN_E_BADARG;
  hipLaunchKernelGGL(bn_bwd_dz_kernel, rowstream_grid(rows, ld), dim3(256), 0, (hipStream_t)stream, dy, z,
                     (const float*)nullptr, 1.0f / (1.0f - p), mean, invstd, gamma, beta, relu, res, dbeta, dgamma, dz,
                     dres, rows, cols, ld, make_drop(p, seed, step_dev, layer));
  return (int)hipGetLastError();
}
// the keep mask the *_drop_* kernels draw, written out (tests; tools): mask[rows][ld] of 0 / 1
__global__ __launch_bounds__(256) void drop_mask_kernel(float4* __restrict__ mask, size_t n4, DropArgs drop) {
  const unsigned dstep = (unsigned)*drop.step;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
    float m[4];
    egn_drop_mask4(drop, dstep, e, m);
    mask[e] = make_float4(m[0], m[1], m[2], m[3]);
  }
}
extern "C" int egn_dropout_mask_f32(float* mask, long n, float p, unsigned long long seed, const int* step_dev, int layer,
                                    void* stream) {
  if (!mask || n <= 0 || n % 4 || !(p > 0.f && p < 1.f) || !step_dev) return EGN_E_BADARG;
  hipLaunchKernelGGL(drop_mask_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<float4*>(mask), (size_t)(n / 4), make_drop(p, seed, step_dev, layer));
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// y = a + b (float4), MSE(mean) loss + gradient, running-stat update, Adam
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                  float4* __restrict__ y, size_t n4) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
    const float4 u = a[e], v = b[e];
    y[e] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}
extern "C" int egn_add_f32(const float* a, const float* b, float* y, long n, void* stream) {
  if (n <= 0 || n % 4) return EGN_E_BADARG;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                     reinterpret_cast<float4*>(y), (size_t)(n / 4));
  return (int)hipGetLastError();
}

// loss[0] += weight * sum((pred - tgt)^2) / (rows*cols)  (the caller zeroes loss);
// dpred (=, or += with accumulate) weight * 2 (pred - tgt) / (rows*cols)
// (MSELoss reduction='mean', function.py:204-215; weight 0.5 = the heat-map term :95-111)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                  int rows, int cols, int ldp, int ldt, float weight,
                                                  int accumulate, float* __restrict__ dpred,
                                                  double* __restrict__ loss) {
  const size_t total = (size_t)rows * cols;
  const double inv = (double)weight / (double)total;
  double acc = 0.0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols), c = (int)(e % cols);
    const float d = pred[(size_t)r * ldp + c] - tgt[(size_t)r * ldt + c];
    acc += (double)d * d;
    if (dpred) {
      const float gd = (float)(2.0 * d * inv);
      float* q = dpred + (size_t)r * ldp + c;
      *q = accumulate ? *q + gd : gd;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv);
}
extern "C" int egn_mse_f32(const float* pred, const float* tgt, int rows, int cols, int ld_pred, int ld_tgt,
                           float weight, int accumulate, float* dpred, double* loss, void* stream) {
  if (rows <= 0 || cols <= 0 || ld_pred < cols || ld_tgt < cols) return EGN_E_BADARG;
  hipLaunchKernelGGL(mse_kernel, dim3(grid_for((size_t)rows * cols, 256) > 256 ? 256 : grid_for((size_t)rows * cols, 256)),
                     dim3(256), 0, (hipStream_t)stream, pred, tgt, rows, cols, ld_pred, ld_tgt, weight, accumulate, dpred, loss);
  return (int)hipGetLastError();
}

// The three criteria of the reference's loss_dict (function.py:17-20) over a [rows, cols] view with row
// pitches: crit 0 = MSELoss, 1 = L1Loss, 2 = SmoothL1Loss (beta = 1), all reduction='mean':
//   loss[0] += weight * mean(c(pred - tgt));   dpred (= or +=) weight * c'(pred - tgt) / (rows*cols)
__global__ __launch_bounds__(256) void elem_loss_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        int rows, int cols, int ldp, int ldt, int crit, float weight,
                                                        int accumulate, float* __restrict__ dpred,
                                                        double* __restrict__ loss) {
  const size_t total = (size_t)rows * cols;
  const double inv = (double)weight / (double)total;
  double acc = 0.0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols), c = (int)(e % cols);
    const float d = pred[(size_t)r * ldp + c] - tgt[(size_t)r * ldt + c];
    const float ad = fabsf(d);
    double v, gd;
    if (crit == 0) { v = (double)d * d; gd = 2.0 * d; }
    else if (crit == 1) { v = ad; gd = d > 0.f ? 1.0 : (d < 0.f ? -1.0 : 0.0); }
    else if (ad < 1.f) { v = 0.5 * (double)d * d; gd = d; }
    else { v = (double)ad - 0.5; gd = d > 0.f ? 1.0 : -1.0; }
    acc += v;
    if (dpred) {
      float* q = dpred + (size_t)r * ldp + c;
      const float gf = (float)(gd * inv);
      *q = accumulate ? *q + gf : gf;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv);
}
extern "C" int egn_elem_loss_f32(const float* pred, const float* tgt, int rows, int cols, int ld_pred, int ld_tgt,
                                 int crit, float weight, int accumulate, float* dpred, double* loss, void* stream) {
  if (rows <= 0 || cols <= 0 || ld_pred < cols || ld_tgt < cols || crit < 0 || crit > 2 || !loss) return EGN_E_BADARG;
  int g = grid_for((size_t)rows * cols, 256);
  if (g > 256) g = 256;
  hipLaunchKernelGGL(elem_loss_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, pred, tgt, rows, cols, ld_pred,
                     ld_tgt, crit, weight, accumulate, dpred, loss);
  return (int)hipGetLastError();
}

// running = (1 - momentum) * running + momentum * batch   (torch BatchNorm semantics)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ running, const float* __restrict__ batch,
                                                  float momentum, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) running[i] = (1.f - momentum) * running[i] + momentum * batch[i];
}
extern "C" int egn_ema_f32(float* running, const float* batch, float momentum, int n, void* stream) {
  if (n <= 0) return EGN_E_BADARG;
  hipLaunchKernelGGL(ema_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, running, batch, momentum, n);
  return (int)hipGetLastError();
}

// torch.optim.Adam (no amsgrad, weight_decay 0): step counted from 1
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[e];
    const float mi = b1 * m[e] + (1.f - b1) * gi;
    const float vi = b2 * v[e] + (1.f - b2) * gi * gi;
    m[e] = mi;
    v[e] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[e] -= (lr / bc1) * (mi / denom);
  }
}
extern "C" int egn_adam_step_f32(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                                 float beta2, float eps, int step, void* stream) {
  if (n <= 0 || step < 1) return EGN_E_BADARG;
  // bias corrections in double on the host, as torch's Python-scalar path does
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2_sqrt = sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (size_t)n, lr, beta1, beta2, eps, (float)bc1, (float)bc2_sqrt);
  return (int)hipGetLastError();
}

// The same update with the step counter and the learning rate in DEVICE memory:
// state[0] = step count (incremented here, by one thread, before the update),
// hyper[0] = lr.  Nothing of the iteration is baked into kernel arguments, so a
// captured hipGraph of a whole training step replays correctly step after step.
__global__ void adam_tick_kernel(int* __restrict__ state) { state[0] += 1; }

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, size_t n,
                                                       const float* __restrict__ hyper, float b1, float b2, float eps,
                                                       const int* __restrict__ state) {
  __shared__ float s_step, s_bc2;
  if (threadIdx.x == 0) {
    const double t = (double)state[0];
    s_step = (float)((double)hyper[0] / (1.0 - pow((double)b1, t)));
    s_bc2 = (float)sqrt(1.0 - pow((double)b2, t));
  }
  __syncthreads();
  const float step_size = s_step, bc2_sqrt = s_bc2;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[e];
    const float mi = b1 * m[e] + (1.f - b1) * gi;
    const float vi = b2 * v[e] + (1.f - b2) * gi * gi;
    m[e] = mi;
    v[e] = vi;
    p[e] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}
// Start-of-step bookkeeping in ONE launch [round 6]: the BatchNorm layers' num_batches_tracked counters (int64, one per
// layer: `counters` is a device array of their addresses) += 1 and the loss accumulator = 0 -- what the steps did with a
// torch._foreach_add_ and a tensor.zero_() (two ATen launches on the critical path of a 1.4 ms step).
// Reference: nn.BatchNorm*d.forward in train mode (libs/model/FCmodel.py:24-43, heatmapModel/hrnet.py:63-92).
__global__ __launch_bounds__(256) void step_counters_kernel(long long* const* __restrict__ counters, int n,
                                                            double* __restrict__ zero_f64) {
  for (int k = threadIdx.x; k < n; k += 256) *counters[k] += 1;
  if (zero_f64 && threadIdx.x == 0) *zero_f64 = 0.0;
}
extern "C" int egn_step_counters_i64(long long* const* counters, int n, double* zero_f64, void* stream) {
  if (n < 0 || (n > 0 && !counters)) return EGN_E_BADARG;
  if (n == 0 && !zero_f64) return 0;
  hipLaunchKernelGGL(step_counters_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counters, n, zero_f64);
  return (int)hipGetLastError();
}

extern "C" int egn_adam_step_dev_f32(float* p, const float* g, float* m, float* v, long n, const float* lr_dev,
                                     float beta1, float beta2, float eps, int* step_dev, void* stream) {
  if (n <= 0 || !lr_dev || !step_dev) return EGN_E_BADARG;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (size_t)n, lr_dev, beta1, beta2, eps, step_dev);
  return (int)hipGetLastError();
}

// torch.optim.Adam with weight_decay (L2, coupled: g += wd * p before the moments, optimizer.py:19-21) and
// torch.optim.SGD(momentum, weight_decay) with dampening 0, no Nesterov (optimizer.py:23-26):
//   g' = g + wd p;   buf = g' on the first step, momentum * buf + g' after;   p -= lr * buf
// Step counter and learning rate in device memory like egn_adam_step_dev_f32.
__global__ __launch_bounds__(256) void adam_l2_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, size_t n,
                                                          const float* __restrict__ hyper, float b1, float b2, float eps,
                                                          float wd, const int* __restrict__ state) {
  __shared__ float s_step, s_bc2;
  if (threadIdx.x == 0) {
    const double t = (double)state[0];
    s_step = (float)((double)hyper[0] / (1.0 - pow((double)b1, t)));
    s_bc2 = (float)sqrt(1.0 - pow((double)b2, t));
  }
  __syncthreads();
  const float step_size = s_step, bc2_sqrt = s_bc2;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float pe = p[e];
    const float gi = g[e] + wd * pe;
    const float mi = b1 * m[e] + (1.f - b1) * gi;
    const float vi = b2 * v[e] + (1.f - b2) * gi * gi;
    m[e] = mi;
    v[e] = vi;
    p[e] = pe - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}
extern "C" int egn_adam_l2_step_dev_f32(float* p, const float* g, float* m, float* v, long n, const float* lr_dev,
                                        float beta1, float beta2, float eps, float weight_decay, int* step_dev,
                                        void* stream) {
  if (n <= 0 || !lr_dev || !step_dev || weight_decay < 0.f) return EGN_E_BADARG;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  hipLaunchKernelGGL(adam_l2_dev_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (size_t)n, lr_dev, beta1, beta2, eps, weight_decay, step_dev);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void sgd_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ buf, size_t n,
                                                      const float* __restrict__ hyper, float momentum, float wd,
                                                      const int* __restrict__ state) {
  const float lr = hyper[0];
  const bool first = state[0] <= 1;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float pe = p[e];
    float gi = g[e] + wd * pe;
    if (momentum != 0.f) {
      gi = first ? gi : momentum * buf[e] + gi;
      buf[e] = gi;
    }
    p[e] = pe - lr * gi;
  }
}
extern "C" int egn_sgd_step_dev_f32(float* p, const float* g, float* buf, long n, const float* lr_dev, float momentum,
                                    float weight_decay, int* step_dev, void* stream) {
  if (n <= 0 || !lr_dev || !step_dev || momentum < 0.f || weight_decay < 0.f || (momentum != 0.f && !buf))
    return EGN_E_BADARG;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  hipLaunchKernelGGL(sgd_dev_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf,
                     (size_t)n, lr_dev, momentum, weight_decay, step_dev);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// Convolution training support
// ---------------------------------------------------------------------------
// torch conv weight [Cout][Cin][KH][KW] -> the conv kernel's packed layout
// [chunk][tap][quad][CoP][4] (input channel = chunk*16 + quad*4 + r), on the
// device, every step (the weights change).
//   dgrad == 0: the forward filter:   out channel = co, in channel = ci, tap
//   dgrad == 1: the data-gradient filter (autograd's conv_transpose view):
//               out channel = ci, in channel = co, tap rotated by 180 degrees,
//               so  dx = conv(dy [zero-inserted for stride 2], packed, stride 1, pad K-1-p)
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                               int taps, int dgrad, float4* __restrict__ dst,
                                                               int CoP, int nchunk) {
  const size_t total = (size_t)nchunk * taps * 4 * CoP;
  const int n_out = dgrad ? Cin : Cout;
  const int n_in = dgrad ? Cout : Cin;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int o = (int)(e % CoP);
    size_t r_ = e / CoP;
    const int quad = (int)(r_ % 4);
    r_ /= 4;
    const int tap = (int)(r_ % taps);
    const int chunk = (int)(r_ / taps);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (o < n_out) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = chunk * EGN_CK + quad * 4 + r;
        if (i < n_in)
          v[r] = dgrad ? w[((size_t)i * Cin + o) * taps + (taps - 1 - tap)] : w[((size_t)o * Cin + i) * taps + tap];
      }
    }
    dst[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" long egn_packed_weight_floats(int Cout, int Cin, int KH, int KW, int dgrad) {
  const int n_out = dgrad ? Cin : Cout, n_in = dgrad ? Cout : Cin;
  if (n_out <= 0 || n_in <= 0 || KH <= 0 || KW <= 0) return -1;
  return (long)((n_in + EGN_CK - 1) / EGN_CK) * KH * KW * 4 * ((n_out + 15) & ~15) * 4;
}

extern "C" int egn_pack_conv_weight_f32(const float* w, int Cout, int Cin, int KH, int KW, int dgrad, float* dst,
                                        void* stream) {
  if (!w || !dst || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return EGN_E_BADARG;
  const int n_out = dgrad ? Cin : Cout, n_in = dgrad ? Cout : Cin;
  const int CoP = (n_out + 15) & ~15;
  const int nchunk = (n_in + EGN_CK - 1) / EGN_CK;
  const size_t total = (size_t)nchunk * KH * KW * 4 * CoP;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, w, Cout,
                     Cin, KH * KW, dgrad, reinterpret_cast<float4*>(dst), CoP, nchunk);
  return (int)hipGetLastError();
}

// All filters of a model in ONE launch: the training step packs every conv weight twice
// per iteration (forward + data-gradient filter, ~600 small launches for HRNet-W48);
// the weights only change in the optimizer step, so the step packs them all up front.
// descs (device memory): one egn_pack_desc per (weight, direction), `begin` = index of
// its first float4 in the global enumeration, ascending; the kernel finds its
// descriptor by binary search.
struct egn_pack_desc {
  const float* w;
  float* dst;
  int Cout, Cin, taps, dgrad;
  long long begin;  // first float4 of this descriptor
};

__global__ __launch_bounds__(256) void pack_conv_weights_batch_kernel(const egn_pack_desc* __restrict__ descs, int n,
                                                                      long long total) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {  // last descriptor with begin <= e
      const int mid = (lo + hi + 1) >> 1;
      if (descs[mid].begin <= e) lo = mid; else hi = mid - 1;
    }
    const egn_pack_desc d = descs[lo];
    const long long le = e - d.begin;
    if (d.dgrad & 4) {
      // [round 5] F(4x4,3x3) filter in conv_wino4.hip's register-feed layout (wino4_pack.h); one unit = one (co, ci)
      // pair = 36 stores.  A wavefront covers the 16 channels li x 4 input channels kq of one (co sub-tile, k-group):
      // each of its store instructions fills one 1 KB block of the (co-tile, k-group, wave) slab.
      const int dg = d.dgrad & 1;
      const int n_out = dg ? d.Cin : d.Cout, n_in = dg ? d.Cout : d.Cin;
      const int li = (int)(le & 15), kq = (int)((le >> 4) & 3);
      long long r_ = le >> 6;
      const int nt = (int)(r_ % 3);
      r_ /= 3;
      const int h = (int)(r_ % (n_in >> 2));
      const int ct = (int)(r_ / (n_in >> 2));
      if (ct * W4P_CO < n_out) w4p_pack_pair(d.w, d.Cin, dg, ct * W4P_CO + nt * 16 + li, 4 * h + kq, n_in, d.dst);
      continue;
    }
    if (d.dgrad & 2) {
      // Winograd filter U = G g G^T (conv_wino.hip layout [co-tile][chunk][f][quad][48][4]); one unit =
      // one (co-tile, chunk, quad, co) = 4 input channels x 16 frequencies = 16 coalesced float4 stores
      const int dg = d.dgrad & 1;
      const int n_out = dg ? d.Cin : d.Cout, n_in = dg ? d.Cout : d.Cin;
      const int nchunk = n_in / EGN_CK;
      const int cot = egn_wino_cot(n_out);   // 48 or 32 output channels per co-tile (conv_wino.hip)
      const int col = (int)(le % cot);
      long long r_ = le / cot;
      const int quad = (int)(r_ % 4);
      r_ /= 4;
      const int chunk = (int)(r_ % nchunk);
      const int ct = (int)(r_ / nchunk);
      const int o = ct * cot + col;
      float u[16][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = chunk * EGN_CK + quad * 4 + r;
        double g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b)
            g[a][b] = dg ? (double)d.w[((size_t)i * d.Cin + o) * 9 + (2 - a) * 3 + (2 - b)]
                         : (double)d.w[((size_t)o * d.Cin + i) * 9 + a * 3 + b];
        double t[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          t[0][b] = g[0][b];
          t[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
          t[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
          t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          u[a * 4 + 0][r] = (float)t[a][0];
          u[a * 4 + 1][r] = (float)(0.5 * (t[a][0] + t[a][1] + t[a][2]));
          u[a * 4 + 2][r] = (float)(0.5 * (t[a][0] - t[a][1] + t[a][2]));
          u[a * 4 + 3][r] = (float)t[a][2];
        }
      }
      float4* slab = reinterpret_cast<float4*>(d.dst) + (size_t)(ct * nchunk + chunk) * (16 * 4 * cot);
#pragma unroll
      for (int f = 0; f < 16; ++f) slab[(f * 4 + quad) * cot + col] = make_float4(u[f][0], u[f][1], u[f][2], u[f][3]);
      continue;
    }
    const int n_out = d.dgrad ? d.Cin : d.Cout;
    const int n_in = d.dgrad ? d.Cout : d.Cin;
    const int CoP = (n_out + 15) & ~15;
    const int o = (int)(le % CoP);
    long long r_ = le / CoP;
    const int quad = (int)(r_ % 4);
    r_ /= 4;
    const int tap = (int)(r_ % d.taps);
    const int chunk = (int)(r_ / d.taps);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (o < n_out) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = chunk * EGN_CK + quad * 4 + r;
        if (i < n_in)
          v[r] = d.dgrad ? d.w[((size_t)i * d.Cin + o) * d.taps + (d.taps - 1 - tap)]
                         : d.w[((size_t)o * d.Cin + i) * d.taps + tap];
      }
    }
    reinterpret_cast<float4*>(d.dst)[le] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" int egn_pack_desc_bytes(void) { return (int)sizeof(egn_pack_desc); }

extern "C" int egn_pack_conv_weights_batch_f32(const void* descs_dev, int n, long total_float4, void* stream) {
  if (!descs_dev || n <= 0 || total_float4 <= 0) return EGN_E_BADARG;
  size_t g = ((size_t)total_float4 + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(pack_conv_weights_batch_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const egn_pack_desc*>(descs_dev), n, (long long)total_float4);
  return (int)hipGetLastError();
}

// up[n][2y][2x][:] = dy[n][y][x][:], everything else zero (data gradient of a
// stride-2 convolution = stride-1 convolution over the zero-inserted dy)
__global__ __launch_bounds__(256) void zero_insert2_kernel(const float4* __restrict__ dy, float4* __restrict__ up,
                                                           int N, int Ho, int Wo, int H, int W, int cs4) {
  const size_t total = (size_t)N * H * W * cs4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % cs4);
    size_t p = e / cs4;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(x & 1) && !(y & 1) && (y >> 1) < Ho && (x >> 1) < Wo)
      v = dy[(((size_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * cs4 + c4];
    up[e] = v;
  }
}
extern "C" int egn_zero_insert2_f32(const float* dy, float* up, int N, int Ho, int Wo, int H, int W, int cs,
                                    void* stream) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || H < 2 * Ho - 1 || W < 2 * Wo - 1 || cs % 4 || cs <= 0) return EGN_E_BADARG;
  const size_t total = (size_t)N * H * W * (cs / 4);
  hipLaunchKernelGGL(zero_insert2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(dy), reinterpret_cast<float4*>(up), N, Ho, Wo, H, W, cs / 4);
  return (int)hipGetLastError();
}

// backward of one term of the multi-resolution fuse y = relu(sum_j up_j(t_j))
// (hrnet.py:282-300): g[n][h][w][:] = sum over the 2^s x 2^s block of dy * (y > 0)
// (s = 0: the ReLU gate only; y == NULL: no gate)
__global__ __launch_bounds__(256) void fuse_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                       float4* __restrict__ g, int N, int H, int W, int cs4, int s) {
  const int hs = H >> s, ws = W >> s, k = 1 << s;
  const size_t total = (size_t)N * hs * ws * cs4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % cs4);
    size_t p = e / cs4;
    const int x = (int)(p % ws);
    p /= ws;
    const int yy = (int)(p % hs);
    const int n = (int)(p / hs);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy_ = 0; dy_ < k; ++dy_)
      for (int dx_ = 0; dx_ < k; ++dx_) {
        const size_t i = (((size_t)n * H + (yy * k + dy_)) * W + (x * k + dx_)) * cs4 + c4;
        float4 d = dy[i];
        if (y) {
          const float4 o = y[i];
          if (!(o.x > 0.f)) d.x = 0.f;
          if (!(o.y > 0.f)) d.y = 0.f;
          if (!(o.z > 0.f)) d.z = 0.f;
          if (!(o.w > 0.f)) d.w = 0.f;
        }
        acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
      }
    g[e] = acc;
  }
}
extern "C" int egn_fuse_bwd_f32(const float* dy, const float* y, float* g, int N, int H, int W, int cs, int shift,
                                void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || cs % 4 || cs <= 0 || shift < 0 || shift > 5 || (H >> shift) << shift != H ||
      (W >> shift) << shift != W)
    return EGN_E_BADARG;
  const size_t total = (size_t)N * (H >> shift) * (W >> shift) * (cs / 4);
  hipLaunchKernelGGL(fuse_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y),
                     reinterpret_cast<float4*>(g), N, H, W, cs / 4, shift);
  return (int)hipGetLastError();
}

// dz = dy * y * (1 - y)   (backward of the Sigmoid that ends the coordinate head, hrnet.py:461-466)
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          float* __restrict__ dz, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float o = y[e];
    dz[e] = dy[e] * o * (1.f - o);
  }
}
extern "C" int egn_sigmoid_bwd_f32(const float* dy, const float* y, float* dz, long n, void* stream) {
  if (n <= 0) return EGN_E_BADARG;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dz,
                     (size_t)n);
  return (int)hipGetLastError();
}

// loss[0] += weight * mean(|pred - tgt|);  dpred = weight * sign(pred - tgt) / n
// (nn.L1Loss(reduction='mean'), the coordinate term of JointsCompositeLoss, function.py:155-168)
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, size_t n,
                                                 float weight, float* __restrict__ dpred, double* __restrict__ loss) {
  const double inv = (double)weight / (double)n;
  double acc = 0.0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float d = pred[e] - tgt[e];
    acc += fabs((double)d);
    if (dpred) dpred[e] = (float)(d > 0.f ? inv : (d < 0.f ? -inv : 0.0));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv);
}
extern "C" int egn_l1_f32(const float* pred, const float* tgt, long n, float weight, float* dpred, double* loss,
                          void* stream) {
  if (n <= 0 || !loss) return EGN_E_BADARG;
  int g = grid_for((size_t)n, 256);
  if (g > 256) g = 256;
  hipLaunchKernelGGL(l1_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, pred, tgt, (size_t)n, weight, dpred, loss);
  return (int)hipGetLastError();
}
