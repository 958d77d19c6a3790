// decode.hip -- key-point decode from NCHW heat-maps, one 64-lane wavefront per
// (n,k) map.  HBM-bound: every map element is read ONCE -- maps of up to 4096 elements (the 64 x 64
// heat-maps of the network) are fetched with 16-byte loads, all in flight together, and stay in
// registers for both passes (maximum, then the soft-arg-max moments); other sizes take the scalar
// two-pass path.
//
// hard (img_proc.py:608-637): flat arg-max, first index on ties,
//   (idx % W, floor(idx / W)), zeroed where max <= 0.
// soft (img_proc.py:678-707): softmax over the flattened map, then
//   x = sum_w w * sum_h p, y = sum_h h * sum_w p  ==  sum_i p_i * (x_i, y_i);
//   maxvals = raw maximum.
// soft-np (img_proc.py:639-676 soft_arg_max_np): same moments with the weights
//   hm / sum(hm) (the np.clip there acts on a copy that is not used), zeroed where
//   max <= 0.
#include "egn_internal.h"

__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(v, off);
    const int oi = __shfl_xor(i, off);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ hm, int nmaps, int H, int W,
                                                     int mode, float* __restrict__ out_xy,
                                                     float* __restrict__ out_max, int32_t* __restrict__ out_idx) {
  const int lane = threadIdx.x & 63;
  const int map = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (map >= nmaps) return;
  const int hw = H * W;
  const float* __restrict__ p = hm + (size_t)map * hw;

  constexpr int VMAX = 16;  // float4 per lane held in registers: maps up to 64 * 16 * 4 = 4096 elements
  const bool vec = (hw & 3) == 0 && hw <= 64 * VMAX * 4 && ((reinterpret_cast<size_t>(p) & 15) == 0);
  float4 reg[VMAX];
  const int nvec = hw >> 2;

  // pass 1: maximum and its first index
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  if (vec) {
    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int j = 0; j < VMAX; ++j) {
      const int q = lane + 64 * j;
      reg[j] = q < nvec ? p4[q] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
#pragma unroll
    for (int j = 0; j < VMAX; ++j) {  // a lane's indices ascend with j and with the component
      const int i0 = 4 * (lane + 64 * j);
      if (reg[j].x > best) { best = reg[j].x; bidx = i0; }
      if (reg[j].y > best) { best = reg[j].y; bidx = i0 + 1; }
      if (reg[j].z > best) { best = reg[j].z; bidx = i0 + 2; }
      if (reg[j].w > best) { best = reg[j].w; bidx = i0 + 3; }
    }
  } else {
    for (int i = lane; i < hw; i += 64) {
      const float v = p[i];
      if (v > best) { best = v; bidx = i; }
    }
  }
  wave_argmax(best, bidx);
  if (bidx == 0x7fffffff) bidx = 0;  // all -inf / NaN map: numpy returns index 0

  float ox, oy;
  if (mode == 0) {
    ox = (float)(bidx % W);
    oy = floorf((float)bidx / (float)W);
    if (!(best > 0.0f)) { ox = 0.f; oy = 0.f; }
  } else if (vec) {
    float s = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
    for (int j = 0; j < VMAX; ++j) {
      const int q = lane + 64 * j;
      if (q < nvec) {
        const float vv[4] = {reg[j].x, reg[j].y, reg[j].z, reg[j].w};
        const int i0 = 4 * q;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float e = mode == 1 ? __expf(vv[c] - best) : vv[c];
          const int y = (i0 + c) / W;
          const int x = (i0 + c) - y * W;
          s += e;
          sx += e * (float)x;
          sy += e * (float)y;
        }
      }
    }
    s = wave_sum(s);
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    ox = sx / s;
    oy = sy / s;
    if (mode == 2 && !(best > 0.0f)) { ox = 0.f; oy = 0.f; }
  } else {
    float s = 0.f, sx = 0.f, sy = 0.f;
    for (int i = lane; i < hw; i += 64) {
      const float e = mode == 1 ? __expf(p[i] - best) : p[i];
      const int y = i / W;
      const int x = i - y * W;
      s += e;
      sx += e * (float)x;
      sy += e * (float)y;
    }
    s = wave_sum(s);
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    ox = sx / s;
    oy = sy / s;
    if (mode == 2 && !(best > 0.0f)) { ox = 0.f; oy = 0.f; }
  }
  if (lane == 0) {
    out_xy[2 * (size_t)map] = ox;
    out_xy[2 * (size_t)map + 1] = oy;
    out_max[map] = best;
    if (out_idx) out_idx[map] = bidx;
  }
}

extern "C" int egn_decode_heatmaps_f32(const float* hm, int N, int K, int H, int W, int mode,
                                       float* out_xy, float* out_max, int32_t* out_idx, void* stream) {
  if (N < 0 || K <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return EGN_E_BADARG;
  const int nmaps = N * K;
  if (nmaps == 0) return 0;
  const int waves_per_block = 4;
  const int grid = (nmaps + waves_per_block - 1) / waves_per_block;
  hipLaunchKernelGGL(decode_kernel, dim3(grid), dim3(64 * waves_per_block), 0, (hipStream_t)stream,
                     hm, nmaps, H, W, mode, out_xy, out_max, out_idx);
  return (int)hipGetLastError();
}
