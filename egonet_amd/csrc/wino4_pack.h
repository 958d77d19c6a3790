// wino4_pack.h -- the F(4x4,3x3) filter transform of ONE (output channel, input channel) pair into conv_wino4.hip's
// register-feed layout, shared by the stand-alone packer (conv_wino4.hip: egn_wino4_pack_weight_f32) and the training
// step's one-launch packer of all filters (train_ops.hip: egn_pack_conv_weights_batch_f32, descriptor code bit 2).
// Reference: the filters of the 3x3 stride-1 convolutions of libs/model/heatmapModel/hrnet.py:63-92.
#pragma once

constexpr int W4P_CO = 48;                    // output channels per co-tile (conv_wino4.hip: W4_CO)
constexpr int W4P_UKG = 12 * 3 * 64 * 4;      // filter floats of one (co-tile, k-group): [wave][3][64 lanes][4] (W4_UKG)

// U = G g G^T in float64, one rounding to fp32 (what engine.pack_wino4_weight computes on the host).  dgrad = 1: the
// data-gradient filter (in / out channels swapped, taps rotated by 180 degrees).  o / i: output / input channel of the
// PACKED filter (dgrad: o < Cin, i < Cout of the torch weight [Cout][Cin][3][3]).  n_in = input channels of the packed
// filter.  36 stores; the pair with (o % 48) < 16 also zeroes the three padding values of its (wave, lane) slots.
__device__ __forceinline__ void w4p_pack_pair(const float* __restrict__ w, int Cin_w, int dgrad, int o, int i, int n_in,
                                              float* __restrict__ dst) {
  double g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      g[a][b] = dgrad ? (double)w[((size_t)i * Cin_w + o) * 9 + (2 - a) * 3 + (2 - b)]
                      : (double)w[((size_t)o * Cin_w + i) * 9 + a * 3 + b];
  // rows of G (points 0, +-1, +-2, inf): [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
  double t[6][3];   // G g
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const double g0 = g[0][b], g1 = g[1][b], g2 = g[2][b];
    t[0][b] = g0 / 4.0;
    t[1][b] = -(g0 + g1 + g2) / 6.0;
    t[2][b] = -(g0 - g1 + g2) / 6.0;
    t[3][b] = g0 / 24.0 + g1 / 12.0 + g2 / 6.0;
    t[4][b] = g0 / 24.0 - g1 / 12.0 + g2 / 6.0;
    t[5][b] = g2;
  }
  const int ct = o / W4P_CO, nt = (o % W4P_CO) >> 4, li = o & 15;
  const int h = i >> 2, kq = i & 3;
  float* base = dst + ((size_t)ct * (n_in >> 2) + h) * W4P_UKG + (16 * kq + li) * 4;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const double t0 = t[a][0], t1 = t[a][1], t2 = t[a][2];
    const double u[6] = {t0 / 4.0, -(t0 + t1 + t2) / 6.0, -(t0 - t1 + t2) / 6.0, t0 / 24.0 + t1 / 12.0 + t2 / 6.0,
                         t0 / 24.0 - t1 / 12.0 + t2 / 6.0, t2};
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int pt = a * 6 + b, wave = pt / 3, p = 3 * (pt % 3) + nt;       // value p = 4 q + r of (wave, lane)
      base[(size_t)wave * 768 + (p >> 2) * 256 + (p & 3)] = (float)u[b];
    }
  }
  if (nt == 0) {
#pragma unroll
    for (int wave = 0; wave < 12; ++wave) {
      float* pad = base + (size_t)wave * 768 + 2 * 256;
      pad[1] = 0.f; pad[2] = 0.f; pad[3] = 0.f;
    }
  }
}
