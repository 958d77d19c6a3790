// Internal declarations shared by the HIP translation units of libegonet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/egonet_hip.h"

// hipFuncSetAttribute is per device: a flag per device (the reference drives several GPUs from one
// process through nn.DataParallel, one thread each), set on the first launch there
constexpr int EGN_MAX_DEVICES = 64;
static inline bool egn_first_use_on_device(bool* seen) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= EGN_MAX_DEVICES) return true;
  if (seen[d]) return false;
  seen[d] = true;
  return true;
}

#define EGN_CHECK_HIP(expr)                      \
  do {                                           \
    hipError_t _e = (expr);                      \
    if (_e != hipSuccess) return (int)_e;        \
  } while (0)

// conv chunking constants (the packed-weight format depends on them)
constexpr int EGN_CK = 16;          // input channels per K chunk
constexpr int EGN_CKQ = EGN_CK / 4; // float4 planes per chunk

struct ConvArgs {
  const float* x;
  const float* w;
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  int N, H, W, Cin, cs_in;
  int Ho, Wo, Cout, cs_out, CoutP;
  int KH, KW, stride, pad;
  int TH, TW, TNB;   // output tile: TNB images x TH rows x TW cols
  int HH, HW;        // input (halo) tile dims per image
  int npix, npixp;   // halo pixels per tile, rounded up to 16
  int tiles_x, tiles_y;
  int nchunk, taps, tps;  // K chunks, taps = KH*KW, taps per LDS stage
  int act, out_nchw;
  int spix_off;  // LDS offset (float4 units) of the tile-row -> output-pixel table
  // training: per-(tile, wave) partial column sums of the stored outputs, [row][2][Cout] doubles
  // (sum, sum of squares), written by the kernels that support it (conv_wino8_kernel); NULL = off
  double* stats;
  // conv_wino9_kernel: multipliers ceil(2^32 / d) for the item index divisions by nct, tiles_x * tiles_y and
  // tiles_x (0 = the divisor is 1), set by its launcher -- the divisions run on the scalar unit
  unsigned mg_nct, mg_txy, mg_tx;
  // conv_wino4c_kernel<., 2> (input channels of an item split over two blocks): one zeroed word per item pair,
  // egn_conv_ticket_count() of them -- the second block to finish applies the epilogue.  NULL: the launcher zeroes y,
  // both blocks add into it and conv_wino4_finish_kernel applies the epilogue (three launches instead of one).
  unsigned* tickets;
};

struct ConvConfig {
  int id;
  int wm, wn, mt, nt;  // waves in M/N, 16x16 sub-tiles per wave in M/N
  int ai, bi;          // dwordx4 staging loads per lane: halo tile / weights of a stage
  int dma;             // 1 = LDS-DMA double-buffered pipeline (conv_dma.hip)
  int tile_m() const { return wm * mt * 16; }
  int tile_n() const { return wn * nt * 16; }
  int threads() const { return 64 * wm * wn; }
};

// co-tile of the Winograd kernels (conv_wino.hip): 48 where Cout allows, else 32, 0 = not supported
__host__ __device__ inline int egn_wino_cot(int cout) { return cout % 48 == 0 ? 48 : (cout % 32 == 0 ? 32 : 0); }

int egn_conv_launch(const ConvArgs& a, int cfg_id, hipStream_t stream);
int egn_conv_plan(ConvArgs& a, int& cfg_id, size_t& lds_bytes);
const ConvConfig* egn_conv_config(int cfg);
int egn_conv_stats_rows(const ConvArgs& a, int cfg_id);
int egn_conv_ticket_count(const ConvArgs& a, int cfg_id);   // a planned for cfg_id; 0 = the config uses none
