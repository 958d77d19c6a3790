// conv_dma.hip -- fp32-MFMA implicit-GEMM convolution, LDS-DMA pipeline.
//
// Same GEMM view, weight packing, fragment mapping and epilogue as
// conv_mfma.hip; what differs is how a K stage reaches LDS:
//
//   * `buffer_load_dwordx4 ... lds` (LDS-DMA) writes the halo tile and the
//     weight slab of stage s+1 straight into the OTHER half of a
//     double-buffered LDS stage while the MFMA loop runs on stage s: no staging
//     VGPRs, no ds_write pass, ONE barrier per stage;
//   * the LDS image of a DMA is lane-linear (wave base + lane*16 B), so the halo
//     tile is stored pixel-major  sA[p][q]  (q = channel quad) -- a lane quad
//     reads 64 contiguous bytes of one pixel.  A-fragment reads are then 2-way
//     bank conflicted (8 instead of 4 LDS cycles per ds_read_b128), which is
//     noise next to 24..48 MFMAs per tap;
//   * zero padding: both halo buffers are cleared once; lanes whose pixel is
//     padding (the same lanes for every chunk) never issue a DMA;
//   * with the staging registers gone the A/B fragments are double-buffered in
//     registers: the ds_reads of tap t+1 are in flight under the MFMAs of tap t.
#include <stdlib.h>

#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One LDS-DMA instruction: 16 B per lane from buffer `r` at voff (+ SGPR soff)
// to LDS at wave-uniform `dst` + lane*16.  Kept out of the kernel template: a
// call to the target builtin with template-dependent operands makes clang drop
// the host-side stub of the kernel without a diagnostic (ROCm 7.2).
//
// Issued as raw ISA, not through __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler
// knows the builtin writes LDS and puts `s_waitcnt vmcnt(0)` in front of the next
// ds_read -- which here belongs to the OTHER stage buffer -- so the DMA of stage s+1
// would have to land before the MFMA loop of stage s may start (measured: that wait sat
// in front of every stage's first fragment read).  The asm is invisible to the waitcnt
// insertion; completion is enforced by the explicit vmcnt(0) in front of the stage
// barrier, one whole MFMA loop after the issue.
__device__ __forceinline__ void egn_dma16(u32x4 r, float4* dst, unsigned voff, int soff) {
#ifdef EGN_DMA_VIA_BUILTIN  // A/B switch for tools/conv_probe.py builds only (-DEGN_DMA_VIA_BUILTIN)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)(r[1] & 0xffffu) << 32) | r[0]), 0, r[2], r[3]);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, voff, soff, 0, 0);
  return;
#endif
  const unsigned lds = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_t)dst;  // wave-uniform LDS byte address
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds), "v"(voff), "s"(r), "s"(soff)
               : "m0");
}

// ABL != 0 are timing ablations (wrong results by construction, never planned):
//   1 no DMA / no barrier in the K loop, 2 = 1 + fragments read once per stage,
//   3 full K loop but no epilogue
template <int WM, int WN, int MT, int NT, int A_IT, int B_IT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_dma_kernel(ConvArgs a) {
  static_assert(WM * WN == 4, "256-thread workgroups");
  constexpr int NTHREADS = 256;
  constexpr int TN = WN * NT * 16;
  constexpr int CKQ = EGN_CKQ;

  extern __shared__ float4 smem[];
  const int a_slots = a.npixp * CKQ;     // float4 per halo buffer (multiple of 64)
  const int b_slots = a.tps * CKQ * TN;  // float4 per weight buffer (multiple of 64)
  float4* sA = smem;                     // [2][a_slots]
  float4* sB = smem + 2 * a_slots;       // [2][b_slots]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 15;
  const int kq = lane >> 4;

  const int tile = blockIdx.x;
  const int tx = tile % a.tiles_x;
  const int ty = (tile / a.tiles_x) % a.tiles_y;
  const int tb = tile / (a.tiles_x * a.tiles_y);
  const int n_base = tb * a.TNB;
  const int oy0 = ty * a.TH;
  const int ox0 = tx * a.TW;
  const int n0 = blockIdx.y * TN;
  const int tile_px = a.TH * a.TW;

  constexpr unsigned OOB = 0xF0000000u;
  // buffer descriptors as raw words (base, stride 0, bytes, flags) for the inline-asm DMA
  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const unsigned long long waddr = reinterpret_cast<unsigned long long>(a.w);
  const u32x4 rx = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu,
                    (unsigned)((size_t)a.N * a.H * a.W * a.cs_in * 4), 0x00020000u};
  const u32x4 rw = {(unsigned)waddr, (unsigned)(waddr >> 32) & 0xffffu,
                    (unsigned)((size_t)a.nchunk * a.taps * CKQ * a.CoutP * 16), 0x00020000u};

  // halo element e = tid + it*256 -> LDS slot e = pixel p * 4 + quad q
  const int q = tid & 3;
  const int p0 = tid >> 2;
  unsigned aoff[A_IT];  // byte offset of (pixel, quad) in x; OOB = padding / beyond the tile
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int p = p0 + it * 64;
    unsigned off = OOB;
    if (p < a.npix) {
      const int hx = p % a.HW;
      const int r = p / a.HW;
      const int hy = r % a.HH;
      const int b = r / a.HH;
      const int n = n_base + b;
      const int iy = oy0 * a.stride - a.pad + hy;
      const int ix = ox0 * a.stride - a.pad + hx;
      if ((n < a.N) && (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W))
        off = (unsigned)(((n * a.H + iy) * a.W + ix) * a.cs_in + q * 4) * 4u;
    }
    aoff[it] = off;
  }
#define EGN_BVOFF(IT)                                                   \
  ((n0 + ((tid + (IT)*NTHREADS) % TN)) < a.CoutP                        \
       ? (unsigned)((((tid + (IT)*NTHREADS) / TN) * a.CoutP) + n0 + ((tid + (IT)*NTHREADS) % TN)) * 16u \
       : OOB)

  // A-fragment base slot (tap 0,0) of this lane for each 16-row sub-tile
  int pixbase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 16 + li;
    int b = m / tile_px;
    const int rem = m - b * tile_px;
    const int y = rem / a.TW;
    const int x = rem - y * a.TW;
    if (b >= a.TNB) b = 0;  // rows beyond the tile: read anything valid, never stored
    pixbase[mt] = ((b * a.HH + y * a.stride) * a.HW + x * a.stride) * CKQ + kq;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nspc = (a.taps + a.tps - 1) / a.tps;  // stages per chunk
  const int nstages = a.nchunk * nspc;

  // clear both halo buffers once: padding slots are never written afterwards;
  // tile row -> output pixel table for the epilogue
  for (int i = tid; i < 2 * a_slots; i += NTHREADS) sA[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!a.out_nchw) conv_epi_pixels<WM, MT>(a, smem, tid, n_base, oy0, ox0);
  __syncthreads();

// LDS-DMA of stage S: halo tile of a new chunk into sA[chunk & 1], weight slab
// into sB[S & 1].  soffset (SGPR) = chunk channel offset / stage slab offset.
#define EGN_DMA(S)                                                                                   \
  {                                                                                                  \
    const int c_ = (S) / nspc;                                                                       \
    const int g_ = (S) - c_ * nspc;                                                                  \
    if (g_ == 0) {                                                                                   \
      float4* dst_ = sA + (c_ & 1) * a_slots + wave * 64;                                            \
      const bool cok_ = (c_ * EGN_CK + q * 4) < a.cs_in;                                             \
      _Pragma("unroll") for (int it = 0; it < A_IT; ++it) {                                          \
        if (it * 64 < a.npix) {                                                                      \
          if (aoff[it] != OOB && cok_)                                                               \
            egn_dma16(rx, dst_ + it * NTHREADS, aoff[it], c_ * EGN_CK * 4);                          \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
    const int nts_ = min(a.tps, a.taps - g_ * a.tps);                                                \
    const int b_elems_ = nts_ * CKQ * TN;                                                            \
    const int sw_ = (c_ * a.taps + g_ * a.tps) * CKQ * a.CoutP * 16;                                 \
    float4* dstb_ = sB + ((S)&1) * b_slots + wave * 64;                                              \
    _Pragma("unroll") for (int it = 0; it < B_IT; ++it) {                                            \
      if (wave * 64 + it * NTHREADS < b_elems_)                                                      \
        egn_dma16(rw, dstb_ + it * NTHREADS, EGN_BVOFF(it), sw_);                                    \
    }                                                                                                \
  }

#define EGN_LOADF(AF, BF, TT)                                                                        \
  {                                                                                                  \
    const int t_ = t0 + (TT);                                                                        \
    const int ky_ = t_ / a.KW;                                                                       \
    const int kx_ = t_ - ky_ * a.KW;                                                                 \
    const int dslot_ = (ky_ * a.HW + kx_) * CKQ;                                                     \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) AF[mt] = curA[pixbase[mt] + dslot_];           \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                \
        BF[nt] = curB[((TT)*CKQ + kq) * TN + (wn * NT + nt) * 16 + li];                              \
  }

#define EGN_MFMA(AF, BF)                                                                             \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[mt].x, BF[nt].x, acc[mt][nt], 0, 0, 0);    \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[mt].y, BF[nt].y, acc[mt][nt], 0, 0, 0);    \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[mt].z, BF[nt].z, acc[mt][nt], 0, 0, 0);    \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[mt].w, BF[nt].w, acc[mt][nt], 0, 0, 0);    \
  }

// Scheduling hint for one tap: an MFMA occupies its pipe for 32 cycles, the
// wave can issue ~6 other instructions meanwhile.  Ask the scheduler to put one
// DS-read / VALU / SALU instruction (the next tap's fragment reads and their
// address math) behind every MFMA instead of a block of them in front.
// Measured [MI355X r1]: the compiler's own placement is better for the selected tiles (MT*NT = 6:
// 24 MFMAs per 5 reads) -- 96 ch 100.5 -> 95.7 us, 192 ch 94.3 -> 92.0 us, backbone 26.70 -> 26.37 ms
// -- so the hint is off unless the library is built with -DEGN_DMA_INTERLEAVE.
#ifndef EGN_DMA_INTERLEAVE
#define EGN_INTERLEAVE()
#else
#define EGN_INTERLEAVE()                                                             \
  _Pragma("unroll") for (int k_ = 0; k_ < MT * NT * 4; ++k_) {                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* 1 MFMA */                  \
    __builtin_amdgcn_sched_group_barrier(0x106, 1, 0); /* 1 DS read | VALU | SALU */ \
  }
#endif

// MFMA loop of stage S on sA[chunk & 1] / sB[S & 1]; the ds_reads of tap t+1 are
// issued before the MFMAs of tap t (register double buffer afA/afB)
#define EGN_COMPUTE(S)                                                  \
  {                                                                     \
    const int c = (S) / nspc;                                           \
    const int g = (S) - c * nspc;                                       \
    const int t0 = g * a.tps;                                           \
    const int nts = min(a.tps, a.taps - t0);                            \
    const float4* curA = sA + (c & 1) * a_slots;                        \
    const float4* curB = sB + ((S)&1) * b_slots;                        \
    float4 afA[MT], bfA[NT], afB[MT], bfB[NT];                          \
    EGN_LOADF(afA, bfA, 0)                                              \
    for (int tt = 0; tt < nts; tt += 2) {                               \
      if constexpr (ABL == 2) {                                         \
        EGN_MFMA(afA, bfA)                                              \
        if (tt + 1 < nts) EGN_MFMA(afA, bfA)                            \
        continue;                                                       \
      }                                                                 \
      /* branch-free prefetch (clamped tap) so that the ds_reads and the */ \
      /* MFMAs share one scheduling region and can be interleaved        */ \
      EGN_LOADF(afB, bfB, min(tt + 1, nts - 1))                         \
      EGN_MFMA(afA, bfA)                                                \
      EGN_INTERLEAVE()                                                  \
      if (tt + 1 < nts) {                                               \
        EGN_LOADF(afA, bfA, min(tt + 2, nts - 1))                       \
        EGN_MFMA(afB, bfB)                                              \
        EGN_INTERLEAVE()                                                \
      }                                                                 \
    }                                                                   \
  }

  constexpr bool kStage = (ABL != 1 && ABL != 2);
  if constexpr (kStage) EGN_DMA(0)
  for (int s = 0; s + 1 < nstages; ++s) {
    // stage s has landed (vmcnt(0) is part of the barrier while a DMA is in
    // flight) and every wave is done with the buffers stage s+1 overwrites
    if constexpr (kStage) {
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): this lane's share of stage s has landed
      __syncthreads();
      EGN_DMA(s + 1)
    }
    EGN_COMPUTE(s)
  }
  // last stage (peeled): the residual loads of the epilogue go out first and
  // are in flight under its MFMAs
  if constexpr (kStage) {
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
  }
  ConvEpiRegs<MT, NT> er;
  const bool nhwc = !a.out_nchw && ABL != 3;
  if (nhwc) conv_epi_prefetch<WM, WN, MT, NT>(a, smem, tid, n0, er);
  EGN_COMPUTE(nstages - 1)

  if constexpr (ABL == 3) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(acc[mt][nt]));
    return;
  }
  if (nhwc)
    conv_epi_finish<WM, WN, MT, NT>(a, acc, smem, tid, n0, er);
  else
    conv_epi_nchw<WM, WN, MT, NT>(a, acc, tid, n_base, oy0, ox0, n0);
}

template <int WM, int WN, int MT, int NT, int ABL>
static int launch_abl(const ConvArgs& a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];  // per instantiation and device
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<WM, WN, MT, NT, 8, 8, ABL>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  }
  const int tiles_b = (a.N + a.TNB - 1) / a.TNB;
  dim3 grid(a.tiles_x * a.tiles_y * tiles_b, (a.CoutP + WN * NT * 16 - 1) / (WN * NT * 16));
  hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 8, 8, ABL>), grid, dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}

template <int WM, int WN, int MT, int NT>
static int launch_one(const ConvArgs& a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];  // per instantiation and device
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<WM, WN, MT, NT, 8, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  }
  const int tiles_b = (a.N + a.TNB - 1) / a.TNB;
  dim3 grid(a.tiles_x * a.tiles_y * tiles_b, (a.CoutP + WN * NT * 16 - 1) / (WN * NT * 16));
  hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 8, 8>), grid, dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}

// dma family: local ids 1..10 = config ids 11..20 (table in conv_plan.hip)
int egn_conv_launch_dma(const ConvArgs& a, int local_id, size_t lds, hipStream_t stream) {
#ifdef EGN_PROBES
  // timing ablations of the 128x48 / 128x96 tiles (tools/conv_probe.py): probe builds only (-DEGN_PROBES) -- the product
  // library neither contains these kernels nor reads the variable
  static const int abl = getenv("EGN_CONV_ABLATE") ? atoi(getenv("EGN_CONV_ABLATE")) : 0;
  if (abl && (local_id == 2 || local_id == 6)) {
    if (local_id == 2) {
      if (abl == 1) return launch_abl<2, 2, 4, 3, 1>(a, lds, stream);
      if (abl == 2) return launch_abl<2, 2, 4, 3, 2>(a, lds, stream);
      if (abl == 3) return launch_abl<2, 2, 4, 3, 3>(a, lds, stream);
    } else {
      if (abl == 1) return launch_abl<4, 1, 2, 3, 1>(a, lds, stream);
      if (abl == 2) return launch_abl<4, 1, 2, 3, 2>(a, lds, stream);
      if (abl == 3) return launch_abl<4, 1, 2, 3, 3>(a, lds, stream);
    }
  }
#endif
  switch (local_id) {
    case 1: return launch_one<4, 1, 4, 3>(a, lds, stream);
    case 2: return launch_one<2, 2, 4, 3>(a, lds, stream);
    case 3: return launch_one<2, 2, 4, 2>(a, lds, stream);
    case 4: return launch_one<4, 1, 4, 1>(a, lds, stream);
    case 5: return launch_one<4, 1, 4, 2>(a, lds, stream);
    case 6: return launch_one<4, 1, 2, 3>(a, lds, stream);
    case 7: return launch_one<2, 2, 2, 3>(a, lds, stream);
    case 8: return launch_one<2, 2, 2, 2>(a, lds, stream);
    case 9: return launch_one<1, 4, 4, 1>(a, lds, stream);
    case 10: return launch_one<1, 4, 2, 3>(a, lds, stream);
    default: return EGN_E_BADARG;
  }
}
