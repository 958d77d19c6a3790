// conv_wgrad.h -- argument block shared by the weight-gradient kernels (conv_wgrad.hip: direct,
// conv_wgrad_wino.hip: Winograd F(2x2,3x3) form for the 3x3 stride-1 layers)
#pragma once
#include <hip/hip_runtime.h>

struct WgradArgs {
  const float* x;
  const float* dy;
  float* part;  // [nsplit][taps][CoP][CiP]
  int N, H, W, Cin, cs_in;
  int Ho, Wo, Cout, cs_out;
  int KH, KW, stride, pad, taps;
  int TH, TW, TNB, lg_tw, lg_thw;  // output-pixel tile: TNB images x TH x TW (TW, TH*TW powers of two)
  int HH, HWd;                     // halo rows / cols per image
  int TP, NHP;                     // output pixels (multiple of 4) / halo pixels per tile
  int tiles_x, tiles_y, ntiles, tiles_per_split, nsplit;
  int co_tiles, ci_tiles, CoP, CiP;
};

// Winograd form (conv_wgrad_wino.hip).  variant 1: 8 x 16 pixel tiles of one image, 2: 8 x 8 tiles of two
// images.  The planner (conv_wgrad.hip) fills tiles_*, ntiles, tiles_per_split, nsplit, co_tiles, ci_tiles.
constexpr int EGN_WGW_LDS_BYTES = 147456;
int egn_wgrad_wino_launch(const WgradArgs& a, int variant, hipStream_t stream);
