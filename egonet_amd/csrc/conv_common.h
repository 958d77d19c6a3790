// conv_common.h -- device helpers shared by the conv kernel families
// (conv_mfma.hip: register-staged pipeline, conv_dma.hip: LDS-DMA pipeline).
#pragma once
#include "egn_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float egn_act(float v, int act) {
  switch (act) {
    case EGN_ACT_RELU: return fmaxf(v, 0.0f);
    case EGN_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case EGN_ACT_LEAKY: return v > 0.0f ? v : 0.01f * v;
    default: return v;
  }
}

// Epilogue of a 4-wave block.  Lane l of wave (wm, wn) owns
// C[(wm*MT+mt)*16 + 4*(l>>4) + r][n0 + (wn*NT+nt)*16 + (l&15)].
// NHWC: accumulators go through LDS (sC[row][col], row stride TNW+4) so that
// every lane stores / reads residuals as 16-B float4 along the channel axis;
// sPix[m] holds the output pixel index of tile row m.  NCHW (heads, final
// Linear): direct stores.  The caller guarantees that all main-loop LDS reads
// are finished only up to its own wave; the function starts with a barrier.
template <int WM, int WN, int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MT][NT], float4* smem, int tid,
                                              int n_base, int oy0, int ox0, int n0) {
  constexpr int TNW = NT * 16;
  constexpr int TM = WM * MT * 16;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 15;
  const int kq = lane >> 4;
  const int tile_px = a.TH * a.TW;
  const int act = a.act & EGN_ACT_MASK;
  const bool res_after = (a.act & EGN_ACT_RES_AFTER) != 0;
  const int howo = a.Ho * a.Wo;

  if (!a.out_nchw) {
    // ---- NHWC epilogue through LDS: float4 stores along the channel axis ----
    constexpr int SC_LD = TNW + 4;  // floats per sC row (keeps 16-B alignment, spreads banks)
    __syncthreads();                // main-loop LDS reads are done
    float* sC = reinterpret_cast<float*>(smem) + (size_t)wave * (MT * 16) * SC_LD;
    int* sPix = reinterpret_cast<int*>(reinterpret_cast<float*>(smem) + (size_t)4 * (MT * 16) * SC_LD);
    if (tid < TM) {  // output pixel index of tile row m = tid, -1 = outside
      const int m = tid;
      const int b = m / tile_px;
      const int rem = m - b * tile_px;
      const int y = rem / a.TW;
      const int x = rem - y * a.TW;
      const int n = n_base + b;
      const int oy = oy0 + y;
      const int ox = ox0 + x;
      sPix[m] = (b < a.TNB && n < a.N && oy < a.Ho && ox < a.Wo) ? (n * a.Ho + oy) * a.Wo + ox : -1;
    }
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
    constexpr int C4 = TNW / 4;            // float4 per row of the wave's slab
    constexpr int NV = MT * 16 * C4;       // float4 per wave
    const int cbase = n0 + wn * TNW;
    for (int idx = lane; idx < NV; idx += 64) {
      const int row = idx / C4;
      const int c4 = idx - row * C4;
      const int pix = sPix[wm * MT * 16 + row];
      const int co = cbase + c4 * 4;
      if (pix < 0 || co >= a.cs_out) continue;
      float4 v = *reinterpret_cast<const float4*>(&sC[row * SC_LD + c4 * 4]);
      const size_t gidx = (size_t)pix * a.cs_out + co;
      float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.res) rv = *reinterpret_cast<const float4*>(a.res + gidx);
      if (a.res && !res_after) { v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
      v.x = egn_act(v.x, act); v.y = egn_act(v.y, act); v.z = egn_act(v.z, act); v.w = egn_act(v.w, act);
      if (a.res && res_after) { v.x = rv.x + v.x; v.y = rv.y + v.y; v.z = rv.z + v.z; v.w = rv.w + v.w; }
      // keep pad channels zero
      if (co + 0 >= a.Cout) v.x = 0.f;
      if (co + 1 >= a.Cout) v.y = 0.f;
      if (co + 2 >= a.Cout) v.z = 0.f;
      if (co + 3 >= a.Cout) v.w = 0.f;
      *reinterpret_cast<float4*>(a.y + gidx) = v;
    }
    return;
  }

  // ---- NCHW epilogue (heads, final Linear): lane owns rows 4*kq + r and column li
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
