// cross_ratio.hip -- the self-supervised cross-ratio term L_cr of JointsCompositeLoss on
// the device.
//
// Reference: libs/loss/function.py:113-153 (calc_cross_ratio_loss, get_cr_mask) and
// libs/common/img_proc.py:709-720 (appro_cr).  For every sample and every one of the L
// four-point lines (A,B,C,D) = coords[idx[l][0..3]]:
//     cr   = (|AC|^2 |BD|^2) / (|BC|^2 |AD|^2) / target_cr^2
//     mask = 1 if the smallest non-zero pairwise distance of the four points > thres
//     loss = sum(mask * crit(cr, 1)) / sum(mask)             (0 when no line is kept)
// The reference walks samples x lines in Python with one tiny autograd graph per line
// and builds the mask on the host with scipy (a device->host copy per step); here line
// values and the per-point gradients are one launch, the normalisation by sum(mask) and
// the scatter into d(coords) a second, deterministic one (no atomics).
#include "egn_internal.h"

#define CR_WS_PER_LINE 10   // 8 gradient floats (4 points x,y), line loss, mask

__device__ __forceinline__ float cr_len2(float ax, float ay, float bx, float by) {
  const float dx = bx - ax, dy = by - ay;
  return dx * dx + dy * dy;
}

// crit: 0 = mse, 1 = l1, 2 = smooth-l1 (beta 1); returns value, *g = d value / d x
__device__ __forceinline__ float cr_crit(int crit, float x, float* g) {
  const float ax = fabsf(x);
  if (crit == 0) { *g = 2.f * x; return x * x; }
  if (crit == 1) { *g = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); return ax; }
  if (ax < 1.f) { *g = x; return 0.5f * x * x; }
  *g = x > 0.f ? 1.f : -1.f;
  return ax - 0.5f;
}

__global__ __launch_bounds__(256) void cr_lines_kernel(const float* __restrict__ coords, int N, int K,
                                                       const int* __restrict__ idx, int L, float target_sq,
                                                       float thres, int crit, float* __restrict__ ws) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * L) return;
  const int s = e / L, l = e - s * L;
  float px[4], py[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float* p = coords + ((size_t)s * K + idx[l * 4 + q]) * 2;
    px[q] = p[0];
    py[q] = p[1];
  }
  // get_cr_mask: min over the non-zero entries of the 4x4 distance matrix
  float dmin = INFINITY;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      const float d = sqrtf(cr_len2(px[a], py[a], px[b], py[b]));
      if (d != 0.f && d < dmin) dmin = d;
    }
  float* out = ws + (size_t)e * CR_WS_PER_LINE;
  const bool keep = dmin != INFINITY && dmin > thres;
  if (!keep) {
#pragma unroll
    for (int q = 0; q < CR_WS_PER_LINE; ++q) out[q] = 0.f;
    return;
  }
  const float ac = cr_len2(px[0], py[0], px[2], py[2]);
  const float bd = cr_len2(px[1], py[1], px[3], py[3]);
  const float bc = cr_len2(px[1], py[1], px[2], py[2]);
  const float ad = cr_len2(px[0], py[0], px[3], py[3]);
  const float cr = ((ac * bd) / (bc * ad)) / target_sq;
  float g;
  const float v = cr_crit(crit, cr - 1.f, &g);
  // d cr / d|AC|^2 = cr/|AC|^2 ... ; d|PQ|^2 / dQ = 2(Q-P), / dP = -2(Q-P)
  const float gac = 2.f * g * cr / ac, gbd = 2.f * g * cr / bd;
  const float gbc = -2.f * g * cr / bc, gad = -2.f * g * cr / ad;
  const float acx = px[2] - px[0], acy = py[2] - py[0];
  const float bdx = px[3] - px[1], bdy = py[3] - py[1];
  const float bcx = px[2] - px[1], bcy = py[2] - py[1];
  const float adx = px[3] - px[0], ady = py[3] - py[0];
  out[0] = -gac * acx - gad * adx;   // A
  out[1] = -gac * acy - gad * ady;
  out[2] = -gbd * bdx - gbc * bcx;   // B
  out[3] = -gbd * bdy - gbc * bcy;
  out[4] = gac * acx + gbc * bcx;    // C
  out[5] = gac * acy + gbc * bcy;
  out[6] = gbd * bdx + gad * adx;    // D
  out[7] = gbd * bdy + gad * ady;
  out[8] = v;
  out[9] = 1.f;
}

// one block: count = sum(mask), loss += weight * sum(v) / count, then every (sample, joint)
// gathers the gradients of the lines it belongs to, in line order
__global__ __launch_bounds__(256) void cr_apply_kernel(const float* __restrict__ ws, int N, int K,
                                                       const int* __restrict__ idx, int L, float weight,
                                                       float* __restrict__ dcoords, double* __restrict__ loss) {
  __shared__ double s_sum[256];
  __shared__ double s_cnt[256];
  double sum = 0.0, cnt = 0.0;
  for (int e = threadIdx.x; e < N * L; e += 256) {
    sum += (double)ws[(size_t)e * CR_WS_PER_LINE + 8];
    cnt += (double)ws[(size_t)e * CR_WS_PER_LINE + 9];
  }
  s_sum[threadIdx.x] = sum;
  s_cnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
      s_cnt[threadIdx.x] += s_cnt[threadIdx.x + off];
    }
    __syncthreads();
  }
  const double count = s_cnt[0];
  if (count == 0.0) return;                       // function.py:124-125: the term is 0
  const double scale = (double)weight / count;
  if (threadIdx.x == 0) loss[0] += s_sum[0] * scale;
  if (!dcoords) return;
  for (int e = threadIdx.x; e < N * K; e += 256) {
    const int s = e / K, j = e - s * K;
    float gx = 0.f, gy = 0.f;
    for (int l = 0; l < L; ++l)
      for (int q = 0; q < 4; ++q)
        if (idx[l * 4 + q] == j) {
          const float* g = ws + ((size_t)s * L + l) * CR_WS_PER_LINE + 2 * q;
          gx += g[0];
          gy += g[1];
        }
    dcoords[(size_t)e * 2 + 0] += (float)(gx * scale);
    dcoords[(size_t)e * 2 + 1] += (float)(gy * scale);
  }
}

extern "C" long egn_cross_ratio_ws_bytes(int N, int L) {
  if (N <= 0 || L <= 0) return 0;
  return (long)N * L * CR_WS_PER_LINE * (long)sizeof(float);
}

extern "C" int egn_cross_ratio_f32(const float* coords, int N, int K, const int* idx, int L, double target_cr,
                                   float thres, int crit, float weight, float* dcoords, double* loss, float* ws,
                                   void* stream) {
  if (!coords || !idx || !loss || !ws || N <= 0 || K <= 0 || L <= 0 || crit < 0 || crit > 2 || target_cr == 0.0)
    return EGN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int lines = N * L;
  hipLaunchKernelGGL(cr_lines_kernel, dim3((lines + 255) / 256), dim3(256), 0, st, coords, N, K, idx, L,
                     (float)(target_cr * target_cr), thres, crit, ws);
  hipLaunchKernelGGL(cr_apply_kernel, dim3(1), dim3(256), 0, st, ws, N, K, idx, L, weight, dcoords, loss);
  return (int)hipGetLastError();
}
