// conv_pw.hip -- the 1x1 convolutions around the 256-channel tensor of `layer1` as ONE streaming kernel [round 5].
//
// Reference: Bottleneck.forward, libs/model/heatmapModel/hrnet.py:95-133 --
//     out = relu(bn3(conv3(h)) + residual)            1x1, 64 -> 256   (h = the block's 3x3 output)
// and the first line of the NEXT Bottleneck (layer1 is four of them, hrnet.py:325, 512-529)
//     h' = relu(bn1(conv1(out)))                      1x1, 256 -> 64.
// As separate launches (the general conv kernels, 146 + 99 us at 64 crops) the 256-channel tensor `out` -- 268 MB at 64
// crops -- is written once and read back twice (conv1 and the next residual); here a block keeps its 32-pixel tile of
// `out` in LDS between the two GEMMs, so HBM sees: h (16 B/lane LDS-DMA), residual (LDS-DMA into the tile's own
// buffer), `out` (written once, 16 B/lane, from LDS), h' -- every byte of the 256-channel tensors moves once per
// direction.  Both filters live in REGISTERS for the whole kernel (BatchNorm scales folded into them on the host,
// engine.fold_pw): lane (li = l & 15, kq = l >> 4) of wave w holds, for GEMM 1, W3'[co = 64 w + 16 nt + li][16 c + 4 kq
// .. + 3] (c = 0..3, nt = 0..3: 64 registers) and for GEMM 2 W1'[co = 16 w + li][16 c + 4 kq .. + 3] (c = 0..15: 64
// registers) -- float4 pieces of the row-major [Cout][Cin] matrices as they lie, no packing.
//
//   block = 4 waves, 32 pixels per tile, TWO blocks per CU (256 VGPRs each): the blocks are independent, so one's
//   epilogue / barriers run under the other's MFMAs (what the 8-wave lock-step form cannot do);
//   GEMM 1: wave w -> all 32 pixels (2 m-tiles) x channels 64 w .. 64 w + 63 (4 n-tiles): 128 MFMAs, A from the
//           double-buffered LDS image of h (one ds_read_b128 per m-tile and 16 channels = 16 MFMAs);
//   item 1: acc + shift3 (+ residual, read from the R tile where the DMA put it) -> ReLU -> back into the R tile;
//   the R tile goes to HBM as `out` (8 ds_read_b128 + 8 buffer_store_dwordx4 per lane) while
//   GEMM 2: wave w -> 32 pixels x channels 16 w .. 16 w + 15: 128 MFMAs, A = the R tile;  h' = relu(acc + shift1).
//   Every vector-memory wait is vmcnt(0), placed where what it waits for was issued a GEMM earlier.
// LDS images are gemm.hip's "KC" tiles: slabs of [rows][32 channels] (128 B per row), the 16-byte quads of a row XOR-ed
// with (row >> 1) & 7 by the SOURCE address of the lane-linear DMA (fragment reads conflict-free:
// tests/test_gemm_design_cpu.py, tests/test_pw_design_cpu.py).
// FUSED = false: GEMM 1 and its item end alone (the downsample conv of the first block -- no residual, no ReLU -- and
// conv3 of the last block, which no conv1 follows).
#include <stdlib.h>

#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_pw_t;

namespace {

constexpr int PW_C1 = 64, PW_C2 = 256, PW_C3 = 64;   // h -> out -> h'
constexpr int PW_TP = 32;                            // pixels per tile
constexpr int PW_NTH = 256;
constexpr int PW_A_BYTES = PW_TP * PW_C1 * 4;        // 8 KB: 2 slabs of [32 rows][32 ch]
constexpr int PW_R_BYTES = PW_TP * PW_C2 * 4;        // 32 KB: 8 slabs
constexpr int PW_SLAB = PW_TP * 128;                 // bytes of a slab
constexpr int PW_LDS = 2 * PW_A_BYTES + PW_R_BYTES;  // 48 KB
}  // namespace

struct PwArgs {
  const float* h;        // [M][64]
  const float* res;      // [M][256] or null
  const float* w3;       // [256][64], BatchNorm scale folded in
  const float* shift3;   // [256]
  const float* w1;       // [64][256], scale folded in (FUSED)
  const float* shift1;   // [64]
  float* out;            // [M][256]
  float* hn;             // [M][64] (FUSED)
  int M, relu1;
};

__device__ __forceinline__ void pw_dma16(u32x4 r, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(r), "s"(soff)
               : "m0", "memory");
}
// byte offset of (row, quad) inside a slab (gemm.hip: kc_off)
__device__ __forceinline__ unsigned pw_off(int row, int quad) { return (unsigned)row * 128u + (unsigned)((quad ^ ((row >> 1) & 7)) << 4); }

template <bool FUSED>
__global__ __launch_bounds__(PW_NTH, 2) void conv_pw_kernel(PwArgs g) {
  extern __shared__ float4 pw_smem[];
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_pw_t)pw_smem;
  char* smc = reinterpret_cast<char*>(pw_smem);
  const unsigned ldsR = lds0 + 2u * PW_A_BYTES;
  char* sR = smc + 2 * PW_A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const bool has_res = g.res != nullptr;

  const unsigned long long haddr = reinterpret_cast<unsigned long long>(g.h);
  const unsigned long long raddr = reinterpret_cast<unsigned long long>(has_res ? g.res : g.h);
  const u32x4 rh = {(unsigned)haddr, (unsigned)(haddr >> 32) & 0xffffu, (unsigned)((size_t)g.M * PW_C1 * 4), 0x00020000u};
  const u32x4 rr = {(unsigned)raddr, (unsigned)(raddr >> 32) & 0xffffu,
                    (unsigned)((size_t)g.M * (has_res ? PW_C2 : PW_C1) * 4), 0x00020000u};
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (unsigned)((size_t)g.M * PW_C2 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(FUSED ? g.hn : g.out, 0,
                                                                     (unsigned)((size_t)g.M * (FUSED ? PW_C3 : PW_C2) * 4), 0x00020000);

  // ---- the filters: registers for the whole kernel
  f32x4 b1[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      b1[c][nt] = *reinterpret_cast<const f32x4*>(g.w3 + (size_t)(64 * wave + 16 * nt + li) * PW_C1 + 16 * c + 4 * kq);
  f32x4 b2[FUSED ? 16 : 1];
  if constexpr (FUSED) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      b2[c] = *reinterpret_cast<const f32x4*>(g.w1 + (size_t)(16 * wave + li) * PW_C2 + 16 * c + 4 * kq);
  }
  float sh3[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) sh3[nt] = g.shift3[64 * wave + 16 * nt + li];
  const float sh1 = FUSED ? g.shift1[16 * wave + li] : 0.f;
  const float lo1 = g.relu1 ? 0.f : -__builtin_inff();

  // ---- DMA pieces (1 KB = 8 rows x 128 B of one slab; the quad order of a row is made by the source address)
  // h tile: 8 pieces, this wave's: p = 2 wave + j -> slab p >> 2, rows 8 (p & 3) ..
  unsigned avoff[2], aldst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = 2 * wave + j, slab = p >> 2, row = 8 * (p & 3) + (lane >> 3);
    const int quad = (lane & 7) ^ ((row >> 1) & 7);
    avoff[j] = (unsigned)((row * PW_C1 + 32 * slab + 4 * quad) * 4);
    aldst[j] = (unsigned)(p * 1024);
  }
  // residual tile: 32 pieces, this wave's: p = 8 wave + j -> slab 2 wave + (j >> 2), rows 8 (j & 3) ..
  unsigned rvoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 8 * q + (lane >> 3);
    const int quad = (lane & 7) ^ ((row >> 1) & 7);
    rvoff[q] = (unsigned)((row * PW_C2 + 4 * quad) * 4);
  }
  // `out` store: 16-byte slot e = 256 j + tid of the R tile (slab j, row tid >> 3, stored quad tid & 7)
  const unsigned so_lds = (unsigned)tid * 16u;
  const unsigned so_glb = (unsigned)(((tid >> 3) * PW_C2 + 4 * ((tid & 7) ^ (((tid >> 3) >> 1) & 7))) * 4);
  // item end 1: element (row = 16 mt + 4 kq + r, co = 64 wave + 16 nt + li) of the R tile:
  // slab 2 wave + (nt >> 1), quad 4 (nt & 1) + (li >> 2), dword li & 3; the XOR term of the row is (2 kq + (r >> 1)) & 7
  unsigned e1off[2][2];     // [nt & 1][r >> 1], without slab / mt / (r & 1) (immediates)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int quad = (4 * a + (li >> 2)) ^ ((2 * kq + b) & 7);
      e1off[a][b] = (unsigned)((4 * kq + 2 * b) * 128 + quad * 16 + (li & 3) * 4);
    }

  const int ntiles = g.M / PW_TP;
  const int gsz = (int)gridDim.x;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  // prologue: h of the first tile
#pragma unroll
  for (int j = 0; j < 2; ++j) pw_dma16(rh, lds0 + aldst[j], avoff[j], (unsigned)t * (PW_TP * PW_C1 * 4));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int par = 0;
  for (; t < ntiles; t += gsz) {
    // everyone's pieces of h(t) are in LDS (their vmcnt(0) lies before this barrier), everyone is past GEMM 2 of the
    // previous tile and its `out` reads: the R tile and the other h buffer may be overwritten
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned pix0 = (unsigned)t * PW_TP;
    if (has_res) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        pw_dma16(rr, ldsR + (unsigned)((8 * wave + j) * 1024), rvoff[j & 3],
                 pix0 * (PW_C2 * 4) + (unsigned)((2 * wave + (j >> 2)) * 128));
    }
    const int tn = t + gsz;
    if (tn < ntiles) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        pw_dma16(rh, lds0 + (unsigned)((par ^ 1) * PW_A_BYTES) + aldst[j], avoff[j], (unsigned)tn * (PW_TP * PW_C1 * 4));
    }
    // ---- GEMM 1
    f32x4 acc1[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc1[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* sA = smc + par * PW_A_BYTES;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 af[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        af[mt] = *reinterpret_cast<const f32x4*>(sA + (c >> 1) * PW_SLAB + pw_off(16 * mt + li, 4 * (c & 1) + kq));
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][s], b1[c][nt][s], acc1[mt][nt], 0, 0, 0);
    }
    // the residual tile (and h of the next tile) have landed -- this wave's pieces, then everyone's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (has_res) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // ---- item end 1: out = act(acc + shift (+ residual)) into the R tile
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* p = reinterpret_cast<float*>(sR + (2 * wave + (nt >> 1)) * PW_SLAB + mt * (16 * 128) + (r & 1) * 128 +
                                              e1off[nt & 1][r >> 1]);
          float v = acc1[mt][nt][r] + sh3[nt];
          if (has_res) v += *p;
          *p = fmaxf(v, lo1);
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's part of the tile is written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- the tile goes to HBM as `out` ...
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sR + j * PW_SLAB + so_lds);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, so_glb, pix0 * (PW_C2 * 4) + (unsigned)(j * 128), 0);
    }
    if constexpr (FUSED) {
      // ---- ... and is the A operand of GEMM 2
      f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        f32x4 af[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          af[mt] = *reinterpret_cast<const f32x4*>(sR + (c >> 1) * PW_SLAB + pw_off(16 * mt + li, 4 * (c & 1) + kq));
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][s], b2[c][s], acc2[mt], 0, 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = fmaxf(acc2[mt][r] + sh1, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rn,
                                                (unsigned)(((16 * mt + 4 * kq + r) * PW_C3 + 16 * wave + li) * 4),
                                                pix0 * (PW_C3 * 4), 0);
        }
    }
    par ^= 1;
  }
}

template <bool FUSED>
static int pw_launch(const PwArgs& g, hipStream_t st) {
  static bool raised[EGN_MAX_DEVICES];
  static int cus = 0;
  auto k = &conv_pw_kernel<FUSED>;
  if (egn_first_use_on_device(raised))
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS));
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int ntiles = g.M / PW_TP;
  const int grid = ntiles < 2 * cus ? ntiles : 2 * cus;
  hipLaunchKernelGGL(k, dim3(grid), dim3(PW_NTH), PW_LDS, st, g);
  return (int)hipGetLastError();
}

// out[M][256] = act1(h[M][64] . w3^T + shift3 (+ res[M][256]))  and, with w1 != NULL,  hn[M][64] = relu(out . w1^T + shift1).
// w3 [256][64] / w1 [64][256]: the 1x1 filters as torch holds them ([Cout][Cin] row-major) with the folded BatchNorm
// scale multiplied in per output channel.  M % 32 == 0; every pointer 16-byte aligned; tensors below 2 GB.
extern "C" int egn_pw_pair_f32(const float* h, const float* res, const float* w3, const float* shift3, const float* w1,
                               const float* shift1, float* out, float* hn, int M, int relu1, void* stream) {
  if (!h || !w3 || !shift3 || !out || M <= 0 || M % PW_TP) return EGN_E_BADARG;
  if ((w1 != nullptr) != (hn != nullptr) || (w1 && !shift1)) return EGN_E_BADARG;
  if ((double)M * PW_C2 * 4.0 >= 2147483648.0) return EGN_E_BADARG;
  if (res == out || h == hn) return EGN_E_BADARG;    // (tiles of other blocks are in flight: no in-place forms)
  PwArgs g = {};
  g.h = h; g.res = res; g.w3 = w3; g.shift3 = shift3; g.w1 = w1; g.shift1 = shift1; g.out = out; g.hn = hn;
  g.M = M; g.relu1 = relu1;
  return w1 ? pw_launch<true>(g, (hipStream_t)stream) : pw_launch<false>(g, (hipStream_t)stream);
}
