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
// column reductions over the batch: block = 64 columns x 4 row groups... one
// block owns 16 float4 column groups (64 columns) and strides over the rows
// with 16 row lanes; LDS tree over the row lanes.  Sums in fp32 per lane
// (rows/16 terms each), combined in double.
//   mode 0: sum[c]              = sum_r a[r][c]
//   mode 1: stats of z          : mean[c], invstd[c] (biased var + eps), var_unbiased[c]
//   mode 2: BN backward sums    : s1[c] = sum dpre, s2[c] = sum dpre * xhat
//           dpre = dy * (mask? mask*keep_scale : 1) * (pre > 0 or no relu), pre = gamma*xhat + beta
// ---------------------------------------------------------------------------
struct ColArgs {
  const float* a;      // z (mode 1, 2) or the matrix to sum (mode 0)
  const float* dy;     // mode 2
  const float* mask;   // mode 2, optional dropout keep mask (0/1)
  const float* mean;   // mode 2
  const float* invstd; // mode 2
  const float* gamma;  // mode 2
  const float* beta;   // mode 2
  float* out0;         // sum | mean | s1
  float* out1;         // - | invstd | s2
  float* out2;         // - | unbiased var | -
  int rows, cols, ld;
  int mode, relu;
  float eps, keep_scale;
};

// stage 1: grid (cols/64, EGN_COL_SPLITS); block = 64 columns x 4 row lanes over its
// slice of the rows; partial sums (double) go to ws[split][2][cols]
constexpr int EGN_COL_SPLITS = 32;

__global__ __launch_bounds__(256) void colreduce_partial_kernel(ColArgs p, double* __restrict__ ws) {
  __shared__ double red0[4][65];
  __shared__ double red1[4][65];
  const int cl = threadIdx.x & 63;   // column within the block's 64
  const int rl = threadIdx.x >> 6;   // row lane 0..3
  const int c = blockIdx.x * 64 + cl;
  const int rows_per = (p.rows + EGN_COL_SPLITS - 1) / EGN_COL_SPLITS;
  const int r_lo = blockIdx.y * rows_per;
  const int r_hi = min(p.rows, r_lo + rows_per);
  double s0 = 0.0, s1 = 0.0;
  if (c < p.cols) {
    float mean = 0.f, istd = 0.f, g = 0.f, b = 0.f;
    if (p.mode == 2) { mean = p.mean[c]; istd = p.invstd[c]; g = p.gamma[c]; b = p.beta[c]; }
    float f0 = 0.f, f1 = 0.f;
    int cnt = 0;
    for (int r = r_lo + rl; r < r_hi; r += 4) {
      const size_t i = (size_t)r * p.ld + c;
      const float v = p.a[i];
      if (p.mode == 0) {
        f0 += v;
      } else if (p.mode == 1) {
        f0 += v;
        f1 += v * v;
      } else {
        const float xhat = (v - mean) * istd;
        float d = p.dy[i];
        if (p.mask) d *= p.mask[i] * p.keep_scale;
        if (p.relu && !(g * xhat + b > 0.f)) d = 0.f;
        f0 += d;
        f1 += d * xhat;
      }
      if (++cnt == 64) {  // flush the fp32 partials to double every 64 terms
        s0 += f0; s1 += f1; f0 = 0.f; f1 = 0.f; cnt = 0;
      }
    }
    s0 += f0; s1 += f1;
  }
  red0[rl][cl] = s0;
  red1[rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < p.cols) {
    ws[((size_t)blockIdx.y * 2 + 0) * p.cols + c] = red0[0][cl] + red0[1][cl] + red0[2][cl] + red0[3][cl];
    ws[((size_t)blockIdx.y * 2 + 1) * p.cols + c] = red1[0][cl] + red1[1][cl] + red1[2][cl] + red1[3][cl];
  }
}

// stage 2: one thread per column combines the EGN_COL_SPLITS partials and finalises
__global__ __launch_bounds__(256) void colreduce_final_kernel(ColArgs p, const double* __restrict__ ws) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < p.cols) {
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < EGN_COL_SPLITS; ++k) {
      t0 += ws[((size_t)k * 2 + 0) * p.cols + c];
      t1 += ws[((size_t)k * 2 + 1) * p.cols + c];
    }
    if (p.mode == 0) {
      p.out0[c] = (float)t0;
    } else if (p.mode == 1) {
      const double m = t0 / p.rows;
      double var = t1 / p.rows - m * m;
      if (var < 0) var = 0;
      p.out0[c] = (float)m;
      p.out1[c] = (float)(1.0 / sqrt(var + (double)p.eps));
      if (p.out2) p.out2[c] = (float)(p.rows > 1 ? var * p.rows / (p.rows - 1) : var);
    } else {
      p.out0[c] = (float)t0;
      p.out1[c] = (float)t1;
    }
  }
}

// ws: caller-provided scratch of egn_colreduce_ws_bytes(cols) bytes
extern "C" long egn_colreduce_ws_bytes(int cols) { return (long)EGN_COL_SPLITS * 2 * cols * (long)sizeof(double); }

static int launch_col(const ColArgs& p, double* ws, void* stream) {
  if (p.rows <= 0 || p.cols <= 0 || p.ld < p.cols || !ws) return EGN_E_BADARG;
  hipLaunchKernelGGL(colreduce_partial_kernel, dim3((p.cols + 63) / 64, EGN_COL_SPLITS), dim3(256), 0,
                     (hipStream_t)stream, p, ws);
  hipLaunchKernelGGL(colreduce_final_kernel, dim3((p.cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, ws);
  return (int)hipGetLastError();
}

extern "C" int egn_colsum_f32(const float* a, int rows, int cols, int ld, float* sum, void* ws, void* stream) {
  ColArgs p = {};
  p.a = a; p.out0 = sum; p.rows = rows; p.cols = cols; p.ld = ld; p.mode = 0;
  return launch_col(p, (double*)ws, stream);
}

extern "C" int egn_bn_stats_f32(const float* z, int rows, int cols, int ld, float eps, float* mean, float* invstd,
                                float* var_unbiased, void* ws, void* stream) {
  ColArgs p = {};
  p.a = z; p.out0 = mean; p.out1 = invstd; p.out2 = var_unbiased;
  p.rows = rows; p.cols = cols; p.ld = ld; p.mode = 1; p.eps = eps;
  return launch_col(p, (double*)ws, stream);
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
                                                         float* __restrict__ y, int rows, int cols, int ld) {
  const int ld4 = ld / 4;
  const size_t total = (size_t)rows * ld4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % ld4);
    const float4 v = reinterpret_cast<const float4*>(z)[e];
    const float in[4] = {v.x, v.y, v.z, v.w};
    float out[4];
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask) mk = reinterpret_cast<const float4*>(mask)[e];
    const float mka[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c4 * 4 + k;
      float o = 0.f;
      if (c < cols) {
        o = gamma[c] * ((in[k] - mean[c]) * invstd[c]) + beta[c];
        if (relu) o = fmaxf(o, 0.f);
        if (mask) o *= mka[k] * keep_scale;
      }
      out[k] = o;
    }
    reinterpret_cast<float4*>(y)[e] = make_float4(out[0], out[1], out[2], out[3]);
  }
}

extern "C" int egn_bn_act_fwd_f32(const float* z, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, const float* mask, float keep_scale, int relu, float* y,
                                  int rows, int cols, int ld, void* stream) {
  if (rows <= 0 || cols <= 0 || ld % 4 || ld < cols) return EGN_E_BADARG;
  const size_t total = (size_t)rows * (ld / 4);
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, z, mean,
                     invstd, gamma, beta, mask, keep_scale, relu, y, rows, cols, ld);
  return (int)hipGetLastError();
}

extern "C" int egn_bn_bwd_sums_f32(const float* dy, const float* z, const float* mask, float keep_scale,
                                   const float* mean, const float* invstd, const float* gamma, const float* beta,
                                   int relu, int rows, int cols, int ld, float* dbeta, float* dgamma, void* ws,
                                   void* stream) {
  ColArgs p = {};
  p.a = z; p.dy = dy; p.mask = mask; p.keep_scale = keep_scale; p.mean = mean; p.invstd = invstd;
  p.gamma = gamma; p.beta = beta; p.relu = relu; p.out0 = dbeta; p.out1 = dgamma;
  p.rows = rows; p.cols = cols; p.ld = ld; p.mode = 2;
  return launch_col(p, (double*)ws, stream);
}

// dz = gamma * invstd * (dpre - dbeta/rows - xhat * dgamma/rows)   (batch-stat BN backward)
__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                        const float* __restrict__ mask, float keep_scale,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int relu,
                                                        const float* __restrict__ dbeta,
                                                        const float* __restrict__ dgamma, float* __restrict__ dz,
                                                        int rows, int cols, int ld) {
  const int ld4 = ld / 4;
  const size_t total = (size_t)rows * ld4;
  const float inv_rows = 1.0f / (float)rows;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % ld4);
    const float4 zv = reinterpret_cast<const float4*>(z)[e];
    const float4 dv = reinterpret_cast<const float4*>(dy)[e];
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask) mk = reinterpret_cast<const float4*>(mask)[e];
    const float zi[4] = {zv.x, zv.y, zv.z, zv.w};
    const float di[4] = {dv.x, dv.y, dv.z, dv.w};
    const float mi[4] = {mk.x, mk.y, mk.z, mk.w};
    float out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c4 * 4 + k;
      float o = 0.f;
      if (c < cols) {
        const float xhat = (zi[k] - mean[c]) * invstd[c];
        float d = di[k];
        if (mask) d *= mi[k] * keep_scale;
        if (relu && !(gamma[c] * xhat + beta[c] > 0.f)) d = 0.f;
        o = gamma[c] * invstd[c] * (d - dbeta[c] * inv_rows - xhat * dgamma[c] * inv_rows);
      }
      out[k] = o;
    }
    reinterpret_cast<float4*>(dz)[e] = make_float4(out[0], out[1], out[2], out[3]);
  }
}

extern "C" int egn_bn_bwd_dz_f32(const float* dy, const float* z, const float* mask, float keep_scale,
                                 const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 int relu, const float* dbeta, const float* dgamma, float* dz, int rows, int cols,
                                 int ld, void* stream) {
  if (rows <= 0 || cols <= 0 || ld % 4 || ld < cols) return EGN_E_BADARG;
  const size_t total = (size_t)rows * (ld / 4);
  hipLaunchKernelGGL(bn_bwd_dz_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, z, mask,
                     keep_scale, mean, invstd, gamma, beta, relu, dbeta, dgamma, dz, rows, cols, ld);
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

// loss[0] += sum((pred - tgt)^2) / (rows*cols)  (loss must be zeroed by the caller);
// dpred = 2 (pred - tgt) / (rows*cols)   (MSELoss reduction='mean', function.py:204-215)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                  int rows, int cols, int ldp, int ldt, float* __restrict__ dpred,
                                                  double* __restrict__ loss) {
  const size_t total = (size_t)rows * cols;
  const double inv = 1.0 / (double)total;
  double acc = 0.0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols), c = (int)(e % cols);
    const float d = pred[(size_t)r * ldp + c] - tgt[(size_t)r * ldt + c];
    acc += (double)d * d;
    if (dpred) dpred[(size_t)r * ldp + c] = (float)(2.0 * d * inv);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv);
}
extern "C" int egn_mse_f32(const float* pred, const float* tgt, int rows, int cols, int ld_pred, int ld_tgt,
                           float* dpred, double* loss, void* stream) {
  if (rows <= 0 || cols <= 0 || ld_pred < cols || ld_tgt < cols) return EGN_E_BADARG;
  hipLaunchKernelGGL(mse_kernel, dim3(grid_for((size_t)rows * cols, 256) > 256 ? 256 : grid_for((size_t)rows * cols, 256)),
                     dim3(256), 0, (hipStream_t)stream, pred, tgt, rows, cols, ld_pred, ld_tgt, dpred, loss);
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
