// conv_fc.hip -- 1x1 stride-1 convolution as a plain row GEMM, for FEW rows: the fully connected layers of the
// lifter at inference batch sizes (reference libs/model/FCmodel.py:29-52, 92-105: Linear + BatchNorm1d + ReLU
// (+ residual) on [N, C]; the engine records them as 1x1 convs on [N, 1, 1, C]) and the 1x1 convs of HRNet's fuse
// layers on the coarse maps (libs/model/heatmapModel/hrnet.py:236-262: 384 -> 48 / 96 / 192 on 8 x 8 maps ...).
// Rows = N * H * W pixels of the NHWC map.  Config id 79; the tuner takes it where it measures fastest.
//
// Why a kernel of its own: with N = 64 rows the general kernels have 8-16 output tiles for 256 CUs and walk
// K = 1024 serially -- 50 us per 0.13 GFLOP layer, five of them on the tail of every bench step (the lifter runs
// after the decode, nothing overlaps it).  Here the grid is (Cout / 16) x (N / 16) blocks of one 16 x 16 output
// tile each (256 blocks at 64 x 1024), the four waves of a block split K (chunk c -> wave c mod 4), operands go
// straight from global memory into MFMA registers (x rows: float4 per lane along K; the standard packed filter
// [chunk][quad][CoutP][4]: float4 per lane, 256 contiguous bytes per 16 lanes), 8 loads in flight per wave, and the
// four partial tiles are summed through LDS in a fixed order.  Epilogue = the conv kernels': scale / shift /
// residual / activation, 64-byte store segments.  Numerics: fp32 fmaf chains per K quarter, summed w0 + w1 + w2 + w3.
#include "conv_common.h"

__global__ __launch_bounds__(256, 2) void conv_fc_kernel(ConvArgs a) {
  __shared__ f32x4 red[3][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int ntile = a.Cout >> 4;
  const int n0 = (blockIdx.x % ntile) * 16, m0 = (blockIdx.x / ntile) * 16;
  const int nchunk = a.nchunk;
  const int rows = a.N * a.H * a.W;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (unsigned)((size_t)rows * a.cs_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.w), 0, (unsigned)((size_t)nchunk * EGN_CKQ * a.CoutP * 16), 0x00020000);
  // lane (row m0 + li, k lanes 4 kq .. 4 kq + 3 of a chunk) / lane (channel n0 + li, the same k lanes)
  const unsigned xo = (unsigned)(((m0 + li) * a.cs_in + 4 * kq) * 4);
  const unsigned wo = (unsigned)((kq * a.CoutP + n0 + li) * 16);
  const unsigned wchunk = (unsigned)(EGN_CKQ * a.CoutP * 16);

  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  int c = wave;
  for (; c + 12 < nchunk; c += 16) {     // four chunks of this wave per round: 8 loads in flight
    f32x4 xa[4], wb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xa[u] = egn_buf_load16(rx, xo + (unsigned)(c + 4 * u) * 64u);
      wb[u] = egn_buf_load16(rw, wo + (unsigned)(c + 4 * u) * wchunk);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][e], wb[u][e], acc, 0, 0, 0);
  }
  for (; c < nchunk; c += 4) {
    const f32x4 xa = egn_buf_load16(rx, xo + (unsigned)c * 64u);
    const f32x4 wb = egn_buf_load16(rw, wo + (unsigned)c * wchunk);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[e], wb[e], acc, 0, 0, 0);
  }
  if (wave > 0) red[wave - 1][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  acc = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];

  // lane owns rows m0 + 4 kq + r (r = 0..3) of channel n0 + li
  const int act = a.act & EGN_ACT_MASK;
  const bool res_after = (a.act & EGN_ACT_RES_AFTER) != 0;
  const int pitch = a.out_nchw ? a.Cout : a.cs_out;
  const float sc = a.scale[n0 + li], sh = a.shift[n0 + li];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 4 * kq + r;
    if (row >= rows) continue;
    const size_t o = (size_t)row * pitch + n0 + li;
    float v = __builtin_fmaf(acc[r], sc, sh);
    const float rs = a.res ? a.res[o] : 0.f;
    v = res_after ? rs + egn_act(v, act) : egn_act(v + rs, act);
    a.y[o] = v;
  }
}

bool egn_conv_fc_applies(const ConvArgs& a) {
  const bool one = a.H == 1 && a.W == 1;
  return a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.Cin % EGN_CK == 0 && a.cs_in >= a.Cin &&
         a.cs_in % 4 == 0 && a.Cout % 16 == 0 && a.CoutP == a.Cout && (a.out_nchw ? one : a.cs_out >= a.Cout) &&
         a.stats == nullptr && (double)a.N * a.H * a.W * a.cs_in * 4.0 < 2147483648.0;
}

int egn_conv_launch_fc(const ConvArgs& a, hipStream_t stream) {
  if (!egn_conv_fc_applies(a)) return EGN_E_BADARG;
  const int grid = (a.Cout / 16) * ((a.N * a.H * a.W + 15) / 16);
  hipLaunchKernelGGL(conv_fc_kernel, dim3(grid), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}
