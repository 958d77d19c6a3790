// targets.hip -- training-target synthesis on the device.
//
// Reference: libs/common/img_proc.py:347-409 generate_target(): for every visible
// joint a (2*3*sigma+1)^2 un-normalised Gaussian "dot" centred on the joint's
// heat-map cell, clipped to the map; joints whose dot lies completely outside
// get target_weight 0.  The reference builds one map per joint in a Python loop
// per sample on the host; here one thread owns one heat-map pixel, so the maps
// of a whole batch are one launch and the [N,K,H,W] target never crosses PCIe.
#include "egn_internal.h"

// joints [N,K,3] f64 (x, y, ignored) in input-image pixels; vis [N,K] or NULL (= visible)
__global__ __launch_bounds__(256) void gaussian_targets_kernel(const double* __restrict__ joints,
                                                               const float* __restrict__ vis, int N, int K, int H,
                                                               int W, double stride_x, double stride_y, double sigma,
                                                               float* __restrict__ target, float* __restrict__ weight) {
  const size_t total = (size_t)N * K * H * W;
  const double tmp = sigma * 3.0;
  const double size = 2.0 * tmp + 1.0;
  const int glen = (int)ceil(size);               // len(np.arange(0, size, 1))
  const float c0 = (float)floor(size / 2.0);      // x0 = y0 = size // 2
  const float denom = (float)(2.0 * sigma * sigma);
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int px = (int)(e % W);
    const int py = (int)((e / W) % H);
    const size_t nk = e / ((size_t)W * H);
    float w = vis ? vis[nk] : 1.f;
    float v = 0.f;
    if (w > 0.5f) {
      // int() truncates toward zero, as in the reference (img_proc.py:377-378)
      const int mu_x = (int)(joints[nk * 3 + 0] / stride_x + 0.5);
      const int mu_y = (int)(joints[nk * 3 + 1] / stride_y + 0.5);
      const int ulx = (int)(mu_x - tmp), uly = (int)(mu_y - tmp);
      const int brx = (int)(mu_x + tmp + 1), bry = (int)(mu_y + tmp + 1);
      if (ulx >= W || uly >= H || brx < 0 || bry < 0) {
        w = 0.f;                                   // nothing of the dot is in bounds (img_proc.py:382-386)
      } else {
        const int gx = px - ulx, gy = py - uly;
        if (gx >= 0 && gy >= 0 && gx < glen && gy < glen && px < min(brx, W) && py < min(bry, H)) {
          const float dx = (float)gx - c0, dy = (float)gy - c0;
          v = expf(-((dx * dx + dy * dy) / denom));
        }
      }
    }
    target[e] = v;
    if (weight && px == 0 && py == 0) weight[nk] = w;
  }
}

extern "C" int egn_gaussian_targets_f32(const double* joints, const float* vis, int N, int K, int H, int W,
                                        double stride_x, double stride_y, double sigma, float* target,
                                        float* weight, void* stream) {
  if (!joints || !target || N <= 0 || K <= 0 || H <= 0 || W <= 0 || !(stride_x > 0) || !(stride_y > 0) ||
      !(sigma > 0))
    return EGN_E_BADARG;
  const size_t total = (size_t)N * K * H * W;
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(gaussian_targets_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, joints, vis, N, K,
                     H, W, stride_x, stride_y, sigma, target, weight);
  return (int)hipGetLastError();
}
