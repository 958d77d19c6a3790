// Host build (g++) of the float64 math the HIP kernels in geometry.hip run
// (egonet_amd/csrc/pose_math.h), exposed with a C ABI for the CPU test-suite.
// TEST INFRASTRUCTURE: not linked into the product library.
#include "../egonet_amd/csrc/pose_math.h"

extern "C" void harness_pose(const double* pred3d, int n, const double* kpt_x, double fx, double cx,
                             int alpha_mode, double* euler, double* alpha) {
  for (int i = 0; i < n; ++i)
    alpha[i] = egn_pose_solve_one(pred3d + 96 * i, kpt_x ? kpt_x[i] : 0.0, fx, cx, alpha_mode, euler + 3 * i);
}

extern "C" void harness_crop_to_screen(const float* local, int n, int K, double mul_x, double mul_y,
                                       const double* center, const double* scale, int crop_w, int crop_h,
                                       double* screen) {
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < K; ++k) {
      const int e = i * K + k;
      const double u = (double)(float)((double)local[2 * e] * mul_x);
      const double v = (double)(float)((double)local[2 * e + 1] * mul_y);
      egn_crop_to_screen(center[2 * i], center[2 * i + 1], scale[2 * i], crop_w, crop_h, u, v,
                         &screen[2 * e], &screen[2 * e + 1]);
    }
}
