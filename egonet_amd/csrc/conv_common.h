// conv_common.h -- device helpers shared by the conv kernel families
// (conv_mfma.hip: register-staged pipeline, conv_dma.hip: LDS-DMA pipeline).
#pragma once
#include "egn_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned EGN_OOB = 0xF0000000u;  // byte offset beyond any tensor the planner accepts

__device__ __forceinline__ float egn_act(float v, int act) {
  switch (act) {
    case EGN_ACT_RELU: return fmaxf(v, 0.0f);
    case EGN_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case EGN_ACT_LEAKY: return v > 0.0f ? v : 0.01f * v;
    default: return v;
  }
}

// non-template wrappers of the buffer builtins (see conv_dma.hip: a target
// builtin with template-dependent operands makes clang drop the kernel's host stub)
__device__ __forceinline__ f32x4 egn_buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void egn_buf_store16(__amdgpu_buffer_rsrc_t r, unsigned voff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}

// ---------------------------------------------------------------------------
// NHWC epilogue of a 4-wave block, in three steps so that the residual loads
// can be put in flight early:
//
//   conv_epi_pixels    (kernel start)  sPix[m] = output pixel index of tile row m
//   conv_epi_prefetch  (before the last K stage, or right before finish)
//                      per-lane byte offsets of its MT*NT float4 of the output
//                      slab + the residual loads (16 B / lane, all in flight)
//   conv_epi_finish    accumulators -> LDS (sC[row][col], row stride TNW+4) ->
//                      float4 per lane along the channel axis: + residual,
//                      activation, pad-channel zeroing, 16-B buffer stores.
// Residual / output go through raw buffer resources: an invalid element gets
// the OOB offset, which loads 0 and drops the store -- no branches.
// Lane l of wave (wm, wn) owns C[(wm*MT+mt)*16 + 4*(l>>4) + r][n0 + (wn*NT+nt)*16 + (l&15)].
// ---------------------------------------------------------------------------
template <int WM, int MT>
__device__ __forceinline__ void conv_epi_pixels(const ConvArgs& a, float4* smem, int tid, int n_base, int oy0,
                                                int ox0) {
  constexpr int TM = WM * MT * 16;
  int* sPix = reinterpret_cast<int*>(smem + a.spix_off);
  if (tid < TM) {
    const int tile_px = a.TH * a.TW;
    const int b = tid / tile_px;
    const int rem = tid - b * tile_px;
    const int y = rem / a.TW;
    const int x = rem - y * a.TW;
    const int n = n_base + b;
    const int oy = oy0 + y;
    const int ox = ox0 + x;
    sPix[tid] = (b < a.TNB && n < a.N && oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
  }
}

template <int MT, int NT>
struct ConvEpiRegs {
  unsigned voff[MT * NT];  // byte offset of this lane's float4 in y / res, EGN_OOB = skip
  f32x4 rv[MT * NT];       // residual values
};

template <int WM, int WN, int MT, int NT>
__device__ __forceinline__ void conv_epi_prefetch(const ConvArgs& a, const float4* smem, int tid, int n0,
                                                  ConvEpiRegs<MT, NT>& er) {
  constexpr int TNW = NT * 16;
  constexpr int C4 = TNW / 4;  // float4 per row of the wave's slab
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int* sPix = reinterpret_cast<const int*>(smem + a.spix_off);
  const int cbase = n0 + wn * TNW;
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res ? a.res : a.y), 0, (unsigned)((size_t)a.N * a.Ho * a.Wo * a.cs_out * 4), 0x00020000);
#pragma unroll
  for (int it = 0; it < MT * NT; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / C4;
    const int c4 = idx - row * C4;
    const int pix = sPix[wm * MT * 16 + row];
    const int co = cbase + c4 * 4;
    er.voff[it] = (pix >= 0 && co < a.cs_out) ? (unsigned)(pix * a.cs_out + co) * 4u : EGN_OOB;
  }
  if (a.res) {
#pragma unroll
    for (int it = 0; it < MT * NT; ++it) er.rv[it] = egn_buf_load16(rr, er.voff[it]);
  } else {
#pragma unroll
    for (int it = 0; it < MT * NT; ++it) er.rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

template <int WM, int WN, int MT, int NT>
__device__ __forceinline__ void conv_epi_finish(const ConvArgs& a, f32x4 (&acc)[MT][NT], float4* smem, int tid,
                                                int n0, ConvEpiRegs<MT, NT>& er) {
  constexpr int TNW = NT * 16;
  constexpr int SC_LD = TNW + 4;  // floats per sC row (keeps 16-B alignment, spreads banks)
  constexpr int C4 = TNW / 4;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;
  const int li = lane & 15;
  const int kq = lane >> 4;
  const int act = a.act & EGN_ACT_MASK;
  const bool res_after = (a.act & EGN_ACT_RES_AFTER) != 0;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      a.y, 0, (unsigned)((size_t)a.N * a.Ho * a.Wo * a.cs_out * 4), 0x00020000);

  __syncthreads();  // main-loop LDS reads are done
  float* sC = reinterpret_cast<float*>(smem) + (size_t)wave * (MT * 16) * SC_LD;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = n0 + (wn * NT + nt) * 16 + li;
    const bool cok = co < a.CoutP;
    const float sc = cok ? a.scale[co] : 0.f;
    const float sh = cok ? a.shift[co] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sC[(mt * 16 + kq * 4 + r) * SC_LD + nt * 16 + li] = acc[mt][nt][r] * sc + sh;
  }
  __syncthreads();
  const int cbase = n0 + wn * TNW;
#pragma unroll
  for (int it = 0; it < MT * NT; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / C4;
    const int c4 = idx - row * C4;
    const int co = cbase + c4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(&sC[row * SC_LD + c4 * 4]);
    const f32x4 rv = er.rv[it];
    if (!res_after) v += rv;  // rv is 0 without a residual
    v.x = egn_act(v.x, act); v.y = egn_act(v.y, act); v.z = egn_act(v.z, act); v.w = egn_act(v.w, act);
    if (res_after) v = rv + v;
    // keep pad channels zero
    if (co + 0 >= a.Cout) v.x = 0.f;
    if (co + 1 >= a.Cout) v.y = 0.f;
    if (co + 2 >= a.Cout) v.z = 0.f;
    if (co + 3 >= a.Cout) v.w = 0.f;
    egn_buf_store16(ry, er.voff[it], v);
  }
}

// NCHW epilogue (heads, final Linear; no residual): lane owns rows 4*kq + r and
// column li of every 16x16 sub-tile, direct dword stores.
template <int WM, int WN, int MT, int NT>
__device__ __forceinline__ void conv_epi_nchw(const ConvArgs& a, f32x4 (&acc)[MT][NT], int tid, int n_base,
                                              int oy0, int ox0, int n0) {
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 15;
  const int kq = lane >> 4;
  const int tile_px = a.TH * a.TW;
  const int act = a.act & EGN_ACT_MASK;
  const int howo = a.Ho * a.Wo;
  const bool tw4 = (a.TW & 3) == 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m0 = (wm * MT + mt) * 16 + kq * 4;
    int on[4], sp[4];  // image index and oy*Wo+ox of each row, sp < 0 = not stored
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r == 0 || !tw4) {
        const int m = m0 + r;
        const int b = m / tile_px;
        const int rem = m - b * tile_px;
        const int y = rem / a.TW;
        const int x = rem - y * a.TW;
        const int oy = oy0 + y;
        const int ox = ox0 + x;
        on[r] = n_base + b;
        sp[r] = (b < a.TNB && on[r] < a.N && oy < a.Ho && ox < a.Wo) ? oy * a.Wo + ox : -1;
        if (tw4) {  // rows 1..3 follow in x
#pragma unroll
          for (int k = 1; k < 4; ++k) {
            on[k] = on[0];
            sp[k] = (sp[0] >= 0 && ox + k < a.Wo) ? sp[0] + k : -1;
          }
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + (wn * NT + nt) * 16 + li;
      if (co >= a.Cout) continue;
      const float sc = a.scale[co];
      const float sh = a.shift[co];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (sp[r] < 0) continue;
        const size_t idx = ((size_t)on[r] * a.Cout + co) * howo + sp[r];
        a.y[idx] = egn_act(acc[mt][nt][r] * sc + sh, act);
      }
    }
  }
}
