// conv_stem.hip -- the network's first convolution: 3x3, stride 2, pad 1, 3 -> 64 channels on the 256 x 256 crop
// (reference libs/model/heatmapModel/hrnet.py:311-314 `conv1` + `bn1` + ReLU; config id 64).
//
// Why a kernel of its own: with 3 input channels the layer has 27 MACs per output -- 0.16 GFLOP per crop against
// 6.3 MB of activations, AI = 11 FLOP/B: HBM bound.  On the general kernels the 3 channels are padded
// to a 16-channel K chunk, so 81 % of the MFMAs multiply zeros and the layer ran at 265 us / 1.2 TB/s (B = 64)
// -- bound by wasted matrix-pipe time, not by HBM.  Here the K dimension is (tap, channel): ONE
// v_mfma_f32_16x16x4_f32 k-step per filter tap, its four k lanes = the 4 floats of an NHWC4 input pixel
// (channel 3 is the zero pad): 9 k-steps instead of 36, 36 MFMAs per 16 output pixels x 64 channels.
//
//   * persistent 4-wave blocks over 16 x 16 output tiles; the 33 x 33 x 4 input halo of the NEXT tile arrives by
//     LDS-DMA (raw ISA, zero padding = out-of-range lanes) while the current one is consumed: double buffered;
//   * A operand: lane (pixel i = l & 15, kq = l >> 4) reads x[2 oy + ky - 1][2 (ox0 + i) + kx - 1][kq] -- one
//     ds_read_b32 per tap (beside fp32 MFMAs an LDS read costs nothing, profiles/r3_mfma_tax.txt);
//   * B operand: the standard packed filter ([chunk 0][tap][quad 0][co][4]: egn_pack_conv_weight_f32) -- lane
//     (co = l & 15, kq) holds w[tap][co][kq] for the 9 taps x 4 co sub-tiles in 36 registers for the whole kernel;
//   * epilogue from the accumulators: y = act(acc * scale + shift), 16 lanes cover 64 contiguous bytes of a pixel.
// Numerics: the same fp32 fmaf chain per output as the general kernels, in (tap, channel) order.
#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_stem_t;

namespace {
constexpr int ST_T = 16;                  // output tile edge
constexpr int ST_H = 2 * ST_T + 1;        // input halo edge: 33
constexpr int ST_SLOTS = ST_H * ST_H;     // 1089 pixels (16 B each)
constexpr int ST_IT = (ST_SLOTS + 255) / 256;   // DMA instructions per lane and tile: 5 (the last one partial)
constexpr int ST_BUF = ST_IT * 256;       // float4 slots per halo buffer
}  // namespace

__device__ __forceinline__ void stem_dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "m0");
}

__global__ __launch_bounds__(256, 2) void conv_stem_kernel(ConvArgs a) {
  extern __shared__ float4 smem[];   // [2][ST_BUF]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_stem_t)smem;

  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const u32x4 rx = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, (unsigned)((size_t)a.N * a.H * a.W * 16), 0x00020000u};
  const unsigned out_bytes = (unsigned)((size_t)a.N * a.Ho * a.Wo * 64 * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, out_bytes, 0x00020000);

  // the filter: 9 taps x 4 co sub-tiles, one register each, for the whole kernel
  float wf[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[t][nt] = a.w[((size_t)(t * EGN_CKQ) * a.CoutP + nt * 16 + li) * 4 + kq];
  float sc[4], sh[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    sc[nt] = a.scale[nt * 16 + li];
    sh[nt] = a.shift[nt * 16 + li];
  }
  const int act = a.act & EGN_ACT_MASK;
  const float act_lo = act == EGN_ACT_RELU ? 0.f : -__builtin_inff();

  // halo slot e = it * 256 + tid -> (hy, hx) of the 33 x 33 input patch
  int hyx[ST_IT];
#pragma unroll
  for (int it = 0; it < ST_IT; ++it) {
    const int e = it * 256 + tid;
    hyx[it] = e < ST_SLOTS ? (((e / ST_H) << 8) | (e % ST_H)) : -1;
  }
  const int tiles_xy = a.tiles_x * a.tiles_y;
  const int ntile = tiles_xy * a.N;

#define STEM_ISSUE(TILE_, P)                                                                            \
  {                                                                                                     \
    const int n_ = (TILE_) / tiles_xy, r_ = (TILE_)-n_ * tiles_xy;                                      \
    const int ty_ = r_ / a.tiles_x, tx_ = r_ - ty_ * a.tiles_x;                                         \
    const int iy0_ = 2 * ty_ * ST_T - 1, ix0_ = 2 * tx_ * ST_T - 1;                                     \
    _Pragma("unroll") for (int it = 0; it < ST_IT; ++it) {                                              \
      if (it * 256 + wave * 64 < ST_SLOTS) {                                                            \
        const unsigned iy_ = (unsigned)(iy0_ + (hyx[it] >> 8)), ix_ = (unsigned)(ix0_ + (hyx[it] & 255)); \
        const bool in_ = hyx[it] >= 0 && iy_ < (unsigned)a.H && ix_ < (unsigned)a.W;                    \
        stem_dma16(rx, lds0 + (unsigned)(((P)*ST_BUF + it * 256 + wave * 64) * 16),                     \
                   in_ ? (unsigned)(((n_ * a.H + (int)iy_) * a.W + (int)ix_) * 16) : EGN_OOB);          \
      }                                                                                                 \
    }                                                                                                   \
  }

  int tile = blockIdx.x;
  if (tile < ntile) STEM_ISSUE(tile, 0)
  int par = 0;
  for (; tile < ntile; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    // the current halo landed in every wave's share; the previous tile's stores may stay in flight
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nxt < ntile) STEM_ISSUE(nxt, par ^ 1)
    const float* hb = reinterpret_cast<const float*>(smem + par * ST_BUF);
    const int n = tile / tiles_xy, r_ = tile - n * tiles_xy;
    const int ty = r_ / a.tiles_x, tx = r_ - ty * a.tiles_x;
    // wave w: output rows 4w .. 4w+3 of the tile, one m-tile (16 pixels of a row) at a time
#pragma unroll 2
    for (int rr = 0; rr < 4; ++rr) {
      const int oyl = 4 * wave + rr;
      f32x4 acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - 3 * ky;
        const float av = hb[((2 * oyl + ky) * ST_H + 2 * li + kx) * 4 + kq];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wf[t][nt], acc[nt], 0, 0, 0);
      }
      // lane owns pixels ox = 4 kq + r (r = 0..3) of this row, channel nt * 16 + li
      const int oy = ty * ST_T + oyl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ox = tx * ST_T + 4 * kq + r;
        const unsigned vo = (oy < a.Ho && ox < a.Wo) ? (unsigned)(((n * a.Ho + oy) * a.Wo + ox) * 64 + li) * 4u : EGN_OOB;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          float v = fmaxf(acc[nt][r] * sc[nt] + sh[nt], act_lo);
          if (act == EGN_ACT_SIGMOID || act == EGN_ACT_LEAKY) v = egn_act(acc[nt][r] * sc[nt] + sh[nt], act);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, vo, nt * 64, 0);
        }
      }
    }
    // every wave is past its LDS reads before the NEXT iteration's DMA (issued after the barrier above) overwrites
    // this buffer: that DMA is issued one iteration later, behind the barrier -- nothing else to do
    par ^= 1;
  }
#undef STEM_ISSUE
}

bool egn_conv_stem_applies(const ConvArgs& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.Cin <= 4 && a.cs_in == 4 && a.Cout == 64 &&
         a.cs_out == 64 && !a.out_nchw && a.res == nullptr && !(a.act & EGN_ACT_RES_AFTER) && (a.H % 2 == 0) && (a.W % 2 == 0);
}
size_t egn_conv_stem_lds_bytes() { return (size_t)2 * ST_BUF * 16; }

int egn_conv_launch_stem(const ConvArgs& a, size_t lds, hipStream_t stream) {
  if (!egn_conv_stem_applies(a) || a.stats) return EGN_E_BADARG;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int ntile = a.tiles_x * a.tiles_y * a.N;
  const int grid = ntile < 2 * cus ? ntile : 2 * cus;     // two 35 KB blocks per CU
  hipLaunchKernelGGL(conv_stem_kernel, dim3(grid), dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}
