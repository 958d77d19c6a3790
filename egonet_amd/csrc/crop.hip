// crop.hip -- GPU crop front end.
//
// Reference call site: libs/model/egonet.py:68-96 crop_single_instance():
//     trans = get_affine_transform(c, s, 0, (height, width))          img_proc.py:26-64
//     instance = cv2.warpAffine(img, trans, (res0, res1), flags=cv2.INTER_LINEAR)
//     instance = pth_trans(instance)      # ToTensor (/255, HWC->CHW) + Normalize(mean, std)
//                                         # libs/dataset/KITTI/car_instance.py:522-531
// one call per bounding box on the host.  Here: the uint8 image is uploaded once,
// one launch produces the normalised fp32 NCHW crops of ALL its boxes -- what the
// backbone consumes -- so neither the per-box warp nor the 786 KB/crop fp32
// host->device copy exists any more.
//
// cv2 (OpenCV 3.4.2, docs/spec-list.txt) is third party and absent here: PARITY
// UNPINNED.  The kernel restates the published algorithm of cv::warpAffine for
// 8-bit INTER_LINEAR (modules/imgproc/src/imgwarp.cpp): invert the 2x3 matrix in
// double; source coordinates in fixed point with AB_BITS = 10, rounded (half to
// even) per row / per column term, + round_delta 16, shifted to INTER_BITS = 5
// (1/32 pixel); bilinear weights (32-fx)(32-fy).. as 15-bit integers (exactly
// 32x the products, so no table fix-up applies); result (sum + 2^14) >> 15;
// BORDER_CONSTANT 0 per tap.
#include "egn_internal.h"

struct CropArgs {
  const uint8_t* img;  // [H][W][3] RGB, row stride `pitch` bytes
  const double* M;     // [n][6] forward affine (image -> crop), row major 2x3
  const float* mean;   // [3]
  const float* stdv;   // [3]
  float* out;          // [n][3][oh][ow]
  int H, W, pitch, n, oh, ow;
};

__global__ __launch_bounds__(256) void crop_warp_normalize_kernel(CropArgs a) {
  const size_t per = (size_t)a.oh * a.ow;
  const size_t total = (size_t)a.n * per;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(e % a.ow);
    const int y = (int)((e / a.ow) % a.oh);
    const int i = (int)(e / per);
    const double* m = a.M + (size_t)i * 6;
    // dst -> src map (cv::warpAffine without WARP_INVERSE_MAP inverts M)
    double D = m[0] * m[4] - m[1] * m[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double i0 = m[4] * D, i1 = -m[1] * D, i3 = -m[3] * D, i4 = m[0] * D;
    const double i2 = -i0 * m[2] - i1 * m[5], i5 = -i3 * m[2] - i4 * m[5];
    const int X0 = (int)rint((i1 * y + i2) * 1024.0) + 16;
    const int Y0 = (int)rint((i4 * y + i5) * 1024.0) + 16;
    const int X = (X0 + (int)rint(i0 * x * 1024.0)) >> 5;
    const int Y = (Y0 + (int)rint(i3 * x * 1024.0)) >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
    const bool x0 = sx >= 0 && sx < a.W, x1 = sx + 1 >= 0 && sx + 1 < a.W;
    const bool y0 = sy >= 0 && sy < a.H, y1 = sy + 1 >= 0 && sy + 1 < a.H;
    const uint8_t* r0 = a.img + (size_t)(y0 ? sy : 0) * a.pitch;
    const uint8_t* r1 = a.img + (size_t)(y1 ? sy + 1 : 0) * a.pitch;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int p00 = (y0 && x0) ? r0[sx * 3 + c] : 0;
      const int p01 = (y0 && x1) ? r0[(sx + 1) * 3 + c] : 0;
      const int p10 = (y1 && x0) ? r1[sx * 3 + c] : 0;
      const int p11 = (y1 && x1) ? r1[(sx + 1) * 3 + c] : 0;
      const int v = (w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11 + 512) >> 10;
      // ToTensor: float / 255, Normalize: (t - mean) / std, both in fp32
      a.out[((size_t)i * 3 + c) * per + (size_t)y * a.ow + x] = ((float)v / 255.f - a.mean[c]) / a.stdv[c];
    }
  }
}

extern "C" int egn_crop_warp_normalize_u8(const uint8_t* img, int H, int W, int pitch, const double* M, int n,
                                          int out_h, int out_w, const float* mean, const float* stdv, float* out,
                                          void* stream) {
  if (!img || !M || !mean || !stdv || !out || H <= 0 || W <= 0 || pitch < 3 * W || n <= 0 || out_h <= 0 ||
      out_w <= 0)
    return EGN_E_BADARG;
  CropArgs a = {img, M, mean, stdv, out, H, W, pitch, n, out_h, out_w};
  const size_t total = (size_t)n * out_h * out_w;
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(crop_warp_normalize_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
