// geometry.hip -- per-instance float64 glue that the reference runs as Python
// loops on the host (egonet.py:436-453, 469-486, 203-295): crop->screen affine,
// lifter (un)normalisation and the Kabsch pose solve.  One thread per instance
// (or per instance x key-point); these kernels move a few hundred bytes per
// instance and exist to keep the pipeline on the device, not for bandwidth.
// The math lives in pose_math.h (shared with the host unit harness).
#include "egn_internal.h"
#include "pose_math.h"

__global__ __launch_bounds__(256) void kpts_to_screen_kernel(
    const float* __restrict__ local, int n, int K, double mul_x, double mul_y,
    const double* __restrict__ center, const double* __restrict__ scale, int crop_w, int crop_h,
    double* __restrict__ screen, const double* __restrict__ mean_in, const double* __restrict__ std_in,
    float* __restrict__ lifter_in, int ld_in) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * K) return;
  const int inst = e / K;
  const int k = e - inst * K;
  // `local_coord *= resolution` is a float32 in-place multiply in the reference
  const double u = (double)(float)((double)local[2 * e] * mul_x);
  const double v = (double)(float)((double)local[2 * e + 1] * mul_y);
  double X, Y;
  egn_crop_to_screen(center[2 * inst], center[2 * inst + 1], scale[2 * inst], crop_w, crop_h, u, v, &X, &Y);
  const int j = 2 * k;
  screen[(size_t)inst * 2 * K + j] = X;
  screen[(size_t)inst * 2 * K + j + 1] = Y;
  if (lifter_in) {
    lifter_in[(size_t)inst * ld_in + j] = (float)((X - mean_in[j]) / std_in[j]);
    lifter_in[(size_t)inst * ld_in + j + 1] = (float)((Y - mean_in[j + 1]) / std_in[j + 1]);
  }
}

extern "C" int egn_keypoints_to_screen_f64(const float* local, int n, int K, double mul_x, double mul_y,
                                           const double* center, const double* scale, int crop_w,
                                           int crop_h, double* screen, const double* mean_in,
                                           const double* std_in, float* lifter_in, int ld_in,
                                           void* stream) {
  if (n < 0 || K <= 0 || (lifter_in && ld_in < 2 * K)) return EGN_E_BADARG;
  if (n == 0) return 0;
  const int total = n * K;
  hipLaunchKernelGGL(kpts_to_screen_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     local, n, K, mul_x, mul_y, center, scale, crop_w, crop_h, screen, mean_in, std_in,
                     lifter_in, ld_in);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void unnormalize_kernel(const float* __restrict__ y, int n, int D, int ld,
                                                          const double* __restrict__ mean,
                                                          const double* __restrict__ std,
                                                          double* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * D) return;
  const int i = e / D;
  const int d = e - i * D;
  out[e] = (double)y[(size_t)i * ld + d] * std[d] + mean[d];
}

extern "C" int egn_unnormalize_f64(const float* y, int n, int D, int ld, const double* mean_out,
                                   const double* std_out, double* pred3d, void* stream) {
  if (n < 0 || D <= 0 || ld < D) return EGN_E_BADARG;
  if (n == 0) return 0;
  const int total = n * D;
  hipLaunchKernelGGL(unnormalize_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, y, n,
                     D, ld, mean_out, std_out, pred3d);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(64) void pose_solve_kernel(const double* __restrict__ pred3d, int n,
                                                        const double* __restrict__ kpt_x, double fx,
                                                        double cxp, int alpha_mode,
                                                        double* __restrict__ euler,
                                                        double* __restrict__ alpha) {
  const int inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= n) return;
  double e[3];
  const double al = egn_pose_solve_one(pred3d + (size_t)inst * 96, kpt_x ? kpt_x[inst] : 0.0, fx, cxp,
                                       alpha_mode, e);
  euler[3 * inst] = e[0];
  euler[3 * inst + 1] = e[1];
  euler[3 * inst + 2] = e[2];
  alpha[inst] = al;
}

extern "C" int egn_pose_solve_f64(const double* pred3d, int n, const double* kpt_x, double fx, double cx,
                                  int alpha_mode, double* euler, double* alpha, void* stream) {
  if (n < 0 || (alpha_mode != 0 && alpha_mode != 1) || (alpha_mode == 0 && !kpt_x)) return EGN_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(pose_solve_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, pred3d, n,
                     kpt_x, fx, cx, alpha_mode, euler, alpha);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Host twins (plain loops over HOST pointers, the same pose_math.h): the reference's CPU plumbing --
// EgoNet.get_keypoints(is_cuda=False) / get_6d_rep on a CPU model (egonet.py:424-467, 279-295;
// BASELINE config 1) -- needs the glue without a GPU.  Not a fallback of the device path: CUDA
// tensors never come here.
extern "C" int egn_keypoints_to_screen_host_f64(const float* local, int n, int K, double mul_x, double mul_y,
                                                const double* center, const double* scale, int crop_w,
                                                int crop_h, double* screen) {
  if (n < 0 || K <= 0 || (n > 0 && (!local || !center || !scale || !screen))) return EGN_E_BADARG;
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < K; ++k) {
      const size_t e = (size_t)i * K + k;
      const double u = (double)(float)((double)local[2 * e] * mul_x);
      const double v = (double)(float)((double)local[2 * e + 1] * mul_y);
      egn_crop_to_screen(center[2 * i], center[2 * i + 1], scale[2 * i], crop_w, crop_h, u, v, &screen[2 * e],
                         &screen[2 * e + 1]);
    }
  return 0;
}

extern "C" int egn_pose_solve_host_f64(const double* pred3d, int n, const double* kpt_x, double fx, double cx,
                                       int alpha_mode, double* euler, double* alpha) {
  if (n < 0 || (alpha_mode != 0 && alpha_mode != 1) || (alpha_mode == 0 && !kpt_x)) return EGN_E_BADARG;
  if (n > 0 && (!pred3d || !euler || !alpha)) return EGN_E_BADARG;
  for (int i = 0; i < n; ++i)
    alpha[i] = egn_pose_solve_one(pred3d + (size_t)96 * i, kpt_x ? kpt_x[i] : 0.0, fx, cx, alpha_mode, euler + 3 * i);
  return 0;
}
