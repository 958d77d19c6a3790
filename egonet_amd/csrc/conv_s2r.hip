// conv_s2r.hip -- 3x3 stride-2 convolution of the 48-channel branch with the FILTER IN REGISTERS [round 5].
//
// Reference: the strided 3x3 conv + BN (+ ReLU) units of the multi-resolution fuse layers and transitions that START at the
// 48-channel, full-resolution branch (libs/model/heatmapModel/hrnet.py:247-274 _make_fuse_layers, :499-507
// _make_transition_layer): 48 -> 48 / 96 / 192 / 384 channels, 23 of the 46 stride-2 launches of a W48 forward.
// On the general direct kernels (conv_dma / conv_mfma) these layers have the shortest K loop of the network
// (9 taps x 48 channels = 27 steps of 16) and pay a filter stage + a barrier for every one of them: 52-78 TFLOP/s.
// Here (the conv_pw.hip recipe):
//   * block = 3 waves, one 16-column n-tile of a 48-channel co-group each; the wave's slice of the filter -- all 9 taps x 48
//     input channels of its 16 output channels, 108 VGPRs -- is loaded ONCE per block from the DIRECT-packed filter
//     ([chunk][tap][quad][CoutP][4]: a lane's float4 is contiguous, 16 lanes read 256 contiguous bytes) and stays in
//     registers while the block walks over its pixel tiles: no filter traffic, no filter barrier in the K loop;
//   * tile = 16 output pixels (2 rows x 8 columns of one image) = ONE MFMA m-tile; its A operand is gathered "im2col" by
//     the LDS-DMA: 27 pieces of 1 KB = (tap, 16-channel chunk) x [16 output pixels][64 bytes], every lane's source address
//     the tap-shifted input pixel of its output pixel (zero padding = an out-of-range offset: the DMA writes 0); the
//     2.25-fold re-read of the input is served by L2;
//   * 108 MFMAs per wave and tile behind 27 ds_read_b128 (conflict-free: the four 16-byte quads of a row are XOR-ed with
//     2 * ((row >> 3) & 1) by the DMA's source address, tests/test_s2r_design_cpu.py), two barriers per tile, every wait
//     vmcnt(0);
//   * FIVE independent blocks per CU (27 KB of LDS, 128 VGPRs = four waves per SIMD): a block's DMA wait and stores run under
//     the others' MFMAs (EGN_S2R_BPC=<1..5>: fewer, for A/B runs);
//   * epilogue straight from the accumulators: acc * scale + shift, ReLU or none, dword stores (16 lanes = 64 contiguous
//     bytes of a pixel).
// Measured (profiles/r5_s2r_probe*.txt, r5_retune_s2r.log): 48 -> 48 @ 32 x 32 at 64 crops 55 -> 33-38 us, 48 -> 96 70 -> 57-61, one
// crop 23 -> 12 us; 4 or 5 blocks per CU are equally fast; TWO co-groups per block sharing one gathered tile (6 waves, half
// the gather traffic) is 12-17 % SLOWER -- what counts is the number of independent blocks, not the gather bytes.
// Applies to: 3x3, stride 2, pad 1, Cin == 48 (unpadded), Cout % 48 == 0 (unpadded), even input maps, Wo % 8 == 0,
// Ho % 2 == 0, NHWC output, no residual, activation none / ReLU.  Filter kind 0 (egn_pack_conv_weight_f32).
#include <stdlib.h>

#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_s2r_t;

namespace {
constexpr int S2_NW = 3, S2_NTH = 64 * S2_NW;
constexpr int S2_CIN = 48, S2_CG = 48;         // input channels; output channels per block (3 n-tiles)
constexpr int S2_TP = 16, S2_TH = 2, S2_TW = 8;   // output pixels of a tile
constexpr int S2_NSLAB = 27;                   // (tap, 16-channel chunk)
constexpr int S2_LDS = S2_NSLAB * 1024;
}  // namespace

__device__ __forceinline__ void s2r_dma16(u32x4 r, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(r)
               : "m0", "memory");
}

__global__ __launch_bounds__(S2_NTH, 4) void conv_s2r_kernel(ConvArgs a) {
  extern __shared__ float4 s2_smem[];
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_s2r_t)s2_smem;
  const char* smc = reinterpret_cast<const char*>(s2_smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int Co = a.Cout, CoP = a.CoutP;
  const int ncg = Co / S2_CG;
  const int tiles_x = a.Wo / S2_TW, tiles_y = a.Ho / S2_TH;
  const int ntile = a.N * tiles_y * tiles_x;
  const int nitem = ntile * ncg;
  const int gsz = (int)gridDim.x;                 // a multiple of ncg (launcher): the block's co-group is fixed
  const int cg = (int)blockIdx.x % ncg;
  const int co = cg * S2_CG + 16 * wave + li;      // this lane's output channel

  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const u32x4 rx = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, (unsigned)((size_t)a.N * a.H * a.W * S2_CIN * 4),
                    0x00020000u};
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)((size_t)a.N * a.Ho * a.Wo * Co * 4), 0x00020000);

  // ---- the filter slice of this wave: b[tap][chunk], element s = input channel 16 chunk + 4 kq + s of output channel co
  f32x4 b[9][3];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      b[tap][ch] = *reinterpret_cast<const f32x4*>(a.w + ((size_t)((ch * 9 + tap) * 4 + kq) * CoP + co) * 4);
  const float sc = a.scale[co], sh = a.shift[co];
  const float lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();

  // ---- DMA lane -> (output pixel row of the tile, stored 16-byte slot); the slot holds quad = slot ^ 2 ((row >> 3) & 1)
  const int drow = lane >> 2;
  const int dquad = (lane & 3) ^ (((drow >> 3) & 1) << 1);
  // ---- fragment read of this lane: row li, quad kq
  const unsigned frag = (unsigned)(li * 64 + ((kq ^ (((li >> 3) & 1) << 1)) << 4));

  for (int item = blockIdx.x; item < nitem; item += gsz) {
    const int t = item / ncg;
    const int tx = t % tiles_x, tyn = t / tiles_x;
    const int ty = tyn % tiles_y, n = tyn / tiles_y;
    // this DMA lane's output pixel and the top-left input pixel of its 3x3 window
    const int oy = S2_TH * ty + (drow >> 3), ox = S2_TW * tx + (drow & 7);
    const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
    const int pixoff = (((n * a.H + iy0) * a.W + ix0) * S2_CIN + 4 * dquad) * 4;
    // everyone is past the fragment reads of the previous tile: the image may be overwritten
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      // piece p = 9 wave + j = 3 tap + chunk (wave-uniform)
      const int p = 9 * wave + j;
      const int tap = p / 3, ch = p - 3 * tap;
      const int ky = tap / 3, kx = tap - 3 * ky;
      const bool ok = (unsigned)(iy0 + ky) < (unsigned)a.H && (unsigned)(ix0 + kx) < (unsigned)a.W;
      const unsigned voff = ok ? (unsigned)(pixoff + ((ky * a.W + kx) * S2_CIN + 16 * ch) * 4) : EGN_OOB;
      s2r_dma16(rx, lds0 + (unsigned)p * 1024u, voff);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces (and the previous tile's stores)
    __builtin_amdgcn_s_barrier();                          // ... everyone's
    asm volatile("" ::: "memory");
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(smc + (tap * 3 + ch) * 1024 + frag);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], b[tap][ch][s], acc, 0, 0, 0);
      }
    // ---- epilogue: rows 4 kq + r = output pixel (2 ty + (kq >> 1), 8 tx + 4 (kq & 1) + r), channel co
    const unsigned vo = (unsigned)((((n * a.Ho + S2_TH * ty + (kq >> 1)) * a.Wo + S2_TW * tx + 4 * (kq & 1)) * Co + co) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = fmaxf(__builtin_fmaf(acc[r], sc, sh), lo);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, vo, (unsigned)(r * Co * 4), 0);
    }
  }
}

bool egn_conv_s2r_applies(const ConvArgs& a) {
  const int act = a.act & EGN_ACT_MASK;
  return a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.Cin == S2_CIN && a.cs_in == S2_CIN &&
         a.Cout % S2_CG == 0 && a.cs_out == a.Cout && !a.out_nchw && a.res == nullptr && !(a.act & EGN_ACT_RES_AFTER) &&
         (act == EGN_ACT_NONE || act == EGN_ACT_RELU) && a.H % 2 == 0 && a.W % 2 == 0 && a.Ho % S2_TH == 0 &&
         a.Wo % S2_TW == 0 && a.stats == nullptr;
}
size_t egn_conv_s2r_lds_bytes() { return S2_LDS; }

int egn_conv_launch_s2r(const ConvArgs& a, hipStream_t stream) {
  if (!egn_conv_s2r_applies(a)) return EGN_E_BADARG;
  static bool raised[EGN_MAX_DEVICES];
  static int cus = 0;
  if (egn_first_use_on_device(raised))
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2r_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      S2_LDS));
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int ncg = a.Cout / S2_CG;
  const int nitem = a.N * (a.Ho / S2_TH) * (a.Wo / S2_TW) * ncg;
  // five blocks per CU (128 VGPRs: four waves per SIMD; 5 x 27 KB of LDS), a multiple of the co-groups
  static const int bpc = [] { const char* e = getenv("EGN_S2R_BPC"); const int v = e ? atoi(e) : 5; return v >= 1 && v <= 5 ? v : 5; }();
  int grid = bpc * cus / ncg * ncg;
  if (grid <= 0) grid = ncg;
  if (nitem < grid) grid = (nitem + ncg - 1) / ncg * ncg;
  hipLaunchKernelGGL(conv_s2r_kernel, dim3(grid), dim3(S2_NTH), S2_LDS, stream, a);
  return (int)hipGetLastError();
}
