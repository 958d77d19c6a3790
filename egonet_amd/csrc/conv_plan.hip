// conv_plan.hip -- host side of the convolution: tile configurations, the launch
// planner (spatial tile, taps per stage, LDS budget) and the dispatcher.
//
// Two kernel families share the tile shapes:
//   ids  1..10  "staged": global -> registers -> LDS, one LDS buffer, 2 barriers
//               per stage, ~50 KB LDS -> 3 blocks per CU   (conv_mfma.hip)
//   ids 11..30  "dma":    buffer_load ... lds straight into a double-buffered
//               LDS stage, 1 barrier per stage, register double-buffered
//               fragments                                   (conv_dma.hip)
//   ids 31..40  retired: a persistent variant of the dma pipeline that no shape ever selected
//               (removed in round 2; the ids stay reserved so measured tables keep their meaning)
//   ids 41..44  filter-resident persistent kernels for the 48 -> 48 3x3 layers (conv_c48.hip)
//   ids 45..55  fused Winograd F(2x2,3x3) kernels for every 3x3 stride-1 layer (conv_wino.hip)
// The engine's tuner times the candidates on the real shape; cfg 0 = cost model.
#include <stdio.h>

#include "egn_internal.h"

int egn_conv_launch_staged(const ConvArgs& a, int cfg_id, size_t lds, hipStream_t stream);
int egn_conv_launch_dma(const ConvArgs& a, int cfg_id, size_t lds, hipStream_t stream);
int egn_conv_launch_c48(const ConvArgs& a, size_t lds, int waves, hipStream_t stream);
int egn_conv_launch_wino(const ConvArgs& a, size_t lds, int variant, hipStream_t stream);
int egn_conv_launch_stem(const ConvArgs& a, size_t lds, hipStream_t stream);   // conv_stem.hip
bool egn_conv_stem_applies(const ConvArgs& a);
size_t egn_conv_stem_lds_bytes();
int egn_conv_launch_wino4(ConvArgs a, size_t lds, int abl, int geo, hipStream_t stream);          // conv_wino4.hip
bool egn_conv_wino4_applies(const ConvArgs& a, int geo);
size_t egn_conv_wino4_lds_bytes(int geo);
int egn_conv_wino4_tickets(const ConvArgs& a, int geo);
int egn_conv_wino4_stats_rows(const ConvArgs& a, int geo);
int egn_conv_launch_s2r(const ConvArgs& a, hipStream_t stream);                         // conv_s2r.hip
bool egn_conv_s2r_applies(const ConvArgs& a);
size_t egn_conv_s2r_lds_bytes();
int egn_conv_launch_fc(const ConvArgs& a, hipStream_t stream);                          // conv_fc.hip
bool egn_conv_fc_applies(const ConvArgs& a);
size_t egn_conv_wino_lds_bytes(int variant, int cout);
int egn_conv_wino_stats_rows(const ConvArgs& a, int variant);

static const ConvConfig kConfigs[] = {
    // id wm wn mt nt ai bi dma (ai / bi = staging depth in dwordx4 per lane)
    {1, 4, 1, 4, 3, 6, 7, 0},   // 256 x 48   (C = 48 layers)
    {2, 2, 2, 4, 3, 6, 8, 0},   // 128 x 96   (C = 96)
    {3, 2, 2, 4, 2, 8, 8, 0},   // 128 x 64   (C = 64, 192, 256, 384)
    {4, 4, 1, 4, 1, 8, 8, 0},   // 256 x 16
    {5, 4, 1, 4, 2, 8, 8, 0},   // 256 x 32
    {6, 4, 1, 2, 3, 8, 8, 0},   // 128 x 48
    {7, 2, 2, 2, 3, 8, 8, 0},   //  64 x 96
    {8, 2, 2, 2, 2, 8, 8, 0},   //  64 x 64
    {9, 1, 4, 4, 1, 8, 8, 0},   //  64 x 64 (one M strip, N across waves)
    {10, 1, 4, 2, 3, 8, 8, 0},  //  32 x 192
    {11, 4, 1, 4, 3, 8, 8, 1},  // the same tile shapes, LDS-DMA pipeline
    {12, 2, 2, 4, 3, 8, 8, 1},
    {13, 2, 2, 4, 2, 8, 8, 1},
    {14, 4, 1, 4, 1, 8, 8, 1},
    {15, 4, 1, 4, 2, 8, 8, 1},
    {16, 4, 1, 2, 3, 8, 8, 1},
    {17, 2, 2, 2, 3, 8, 8, 1},
    {18, 2, 2, 2, 2, 8, 8, 1},
    {19, 1, 4, 4, 1, 8, 8, 1},
    {20, 1, 4, 2, 3, 8, 8, 1},
    {21, 4, 1, 4, 3, 8, 8, 2},  // LDS-DMA pipeline with a 53 KB LDS budget (3 blocks / CU:
    {22, 2, 2, 4, 3, 8, 8, 2},  // fewer taps per stage, more barriers, more waves to hide
    {23, 2, 2, 4, 2, 8, 8, 2},  // per-block prologue / epilogue)
    {24, 4, 1, 4, 1, 8, 8, 2},
    {25, 4, 1, 4, 2, 8, 8, 2},
    {26, 4, 1, 2, 3, 8, 8, 2},
    {27, 2, 2, 2, 3, 8, 8, 2},
    {28, 2, 2, 2, 2, 8, 8, 2},
    {29, 1, 4, 4, 1, 8, 8, 2},
    {30, 1, 4, 2, 3, 8, 8, 2},
    {31, 4, 1, 4, 3, 8, 8, 3},  // 31..40 retired (dma == 3): never planned, never launched
    {32, 2, 2, 4, 3, 8, 8, 3},
    {33, 2, 2, 4, 2, 8, 8, 3},
    {34, 4, 1, 4, 1, 8, 8, 3},
    {35, 4, 1, 4, 2, 8, 8, 3},
    {36, 4, 1, 2, 3, 8, 8, 3},
    {37, 2, 2, 2, 3, 8, 8, 3},
    {38, 2, 2, 2, 2, 8, 8, 3},
    {39, 1, 4, 4, 1, 8, 8, 3},
    {40, 1, 4, 2, 3, 8, 8, 3},
    {41, 4, 1, 2, 3, 9, 0, 4},  // 48 -> 48 3x3 s1 only: filter resident in LDS, persistent (conv_c48.hip)
    {42, 8, 1, 1, 3, 5, 0, 4},  // the same with 8 waves (two per SIMD)
    {43, 4, 1, 2, 3, 9, 1, 4},  // 4 waves, each with the whole filter in REGISTERS (bi = 1 marks it)
    {44, 8, 1, 2, 3, 3, 2, 4},  // 8 waves x 2 rows on a 16 x 16 tile, halo as a ring of chunks (bi = 2)
    // fused Winograd F(2x2,3x3) (conv_wino.hip): 3x3 s1 p1, Cin % 16 == 0, Cout % 48 == 0.  `w` must be
    // the TRANSFORMED filter (egn_wino_pack_weight_f32 / engine.pack_wino_weight), not the direct pack.
    {45, 4, 1, 1, 3, 6, 0, 5},  // 16 x 16 pixel tile of one image (bi = variant 0)
    {46, 4, 1, 1, 3, 7, 1, 5},  // four 8 x 8 images per block (bi = variant 1)
    {47, 4, 1, 1, 3, 6, 0x10, 5},  // timing ablations of 45 (WRONG RESULTS, tools/wino_probe.py only):
    {48, 4, 1, 1, 3, 6, 0x20, 5},  // no DMA / no input transform / no epilogue memory ops / no barriers
    {49, 4, 1, 1, 3, 6, 0x30, 5},
    {50, 4, 1, 1, 3, 6, 0x40, 5},
    {51, 8, 1, 1, 3, 4, 2, 5},     // the 8-wave Winograd kernel (two per SIMD, frequency halves): 16 x 16 tile
    {52, 8, 1, 1, 3, 4, 3, 5},     // ... four 8 x 8 images
    {53, 8, 1, 1, 3, 4, 0x12, 5},  // timing ablations of 51
    {54, 8, 1, 1, 3, 4, 0x22, 5},
    {55, 8, 1, 1, 3, 4, 0x32, 5},
    {56, 4, 1, 1, 3, 4, 4, 5},     // frequency-halves kernel with 4 waves on two 8 x 8 images (32 tiles per block)
    {57, 4, 1, 1, 3, 4, 5, 5},     // ... on an 8 x 16 pixel tile of one image
    {58, 8, 1, 1, 3, 4, 0x42, 5},  // 51 with s_memtime stamps (tools/wino_clk.py; `res` = the stamp buffer)
    {59, 8, 1, 1, 3, 4, 6, 5},     // conv_wino9_kernel (half the VALU instructions of 51 / 52 / 56 / 57): 16 x 16 tile
    {60, 8, 1, 1, 3, 4, 7, 5},     // ... four 8 x 8 images
    {61, 4, 1, 1, 3, 4, 8, 5},     // ... two 8 x 8 images, 4 waves
    {62, 4, 1, 1, 3, 4, 9, 5},     // ... 8 x 16 pixel tile, 4 waves
    {63, 8, 1, 1, 3, 4, 0x46, 5},  // 59 with s_memtime stamps (tools/wino_clk.py)
    {64, 4, 1, 1, 4, 0, 0, 6},     // the stem: 3x3 s2, 3 -> 64 channels, K = (tap, channel) (conv_stem.hip)
    {65, 6, 1, 1, 3, 4, 10, 5},    // fused Winograd F(4x4,3x3), conv_wino43_kernel: filter from egn ... kind 2
    {66, 6, 1, 1, 3, 4, 0x1a, 5},  // 65 with s_memtime stamps
    {67, 4, 1, 1, 3, 4, 11, 5},    // conv_wino9_kernel with 8-channel stages: 8 x 16 tile, 4 waves, TWO blocks per CU
    {68, 4, 1, 1, 3, 4, 12, 5},    // ... two 8 x 8 images
    {69, 4, 1, 1, 3, 4, 0x4b, 5},  // 67 with s_memtime stamps (tools/wino_clk.py)
    {70, 12, 1, 1, 3, 0, 0, 7},    // fused Winograd F(4x4,3x3), conv_wino4_kernel (conv_wino4.hip): filter kind 3
    {71, 12, 1, 1, 3, 0, 1, 7},    // timing ablations of 70 (WRONG RESULTS, tools/wino_probe.py only): no input transform
    {72, 12, 1, 1, 3, 0, 2, 7},    // ... no MFMAs
    {73, 12, 1, 1, 3, 0, 4, 7},    // ... no exchange / output transform / stores
    {74, 12, 1, 1, 3, 0, 8, 7},    // ... no filter loads
    {75, 12, 1, 1, 3, 0, 16, 7},   // ... no halo DMA
    {76, 12, 1, 1, 3, 0, 7, 7},    // ... only DMA + filter loads + barriers
    {77, 12, 1, 1, 3, 0, 32, 7},   // ... halo reads without bank conflicts
    {78, 12, 1, 1, 3, 0, 64, 7},   // 70 with s_memtime stamps (tools/wino4_clk.py; `res` = the stamp buffer)
    {79, 4, 1, 1, 1, 0, 0, 8},     // 1x1 conv on 1 x 1 maps (the lifter's Linear layers): one 16 x 16 tile per block, K split over the waves (conv_fc.hip)
    {80, 12, 1, 1, 3, 1, 0, 7},    // conv_wino4b_kernel: F(4x4,3x3) on 16 x 16 pixel regions, 16-channel stages (ai = geometry 1); filter kind 3
    {81, 12, 1, 1, 3, 1, 64, 7},   // 80 with s_memtime stamps (tools/wino4_clk.py)
    {82, 12, 1, 1, 3, 2, 0, 7},    // conv_wino4c_kernel<0, 1>: F(4x4,3x3) on 8 x 8 maps, four images per region (ai = geometry 2); filter kind 3
    {83, 12, 1, 1, 3, 6, 0, 7},    // conv_wino4c_kernel<0, 2>: 82 with the input channels of an item split over two blocks (ai bit 2): memset, atomic adds, conv_wino4_finish_kernel
    {84, 12, 1, 1, 3, 5, 0, 7},    // conv_wino4bk_kernel: 80 with the input channels of an item split over two blocks (ai bit 2), as 83
    {85, 3, 1, 1, 3, 0, 0, 9},     // conv_s2r_kernel [round 5]: 3x3 stride 2 from the 48-channel branch, filter slice in registers (conv_s2r.hip); direct-packed filter
    {86, 12, 1, 1, 3, 9, 0, 7},    // conv_wino4w_kernel [round 6]: F(4x4,3x3), 16 x 16 pixel regions x 96 output channels per item (ai bit 3; conv_wino4w.hip); filter kind 3
    {87, 12, 1, 1, 3, 9, 64, 7},   // 86 with s_memtime stamps (tools/wino4_clk.py)
    {88, 6, 1, 1, 3, 17, 0, 7},    // (probe builds only: NEGATIVE result) conv_wino4h_kernel [round 6]: F(4x4,3x3) in half-size blocks (6 waves, 16 tiles x 48 channels, 62 KB), two independent blocks per CU (ai bit 4; conv_wino4h.hip); filter kind 3
    {89, 6, 1, 1, 3, 17, 64, 7},   // 88 with s_memtime stamps (tools/wino4_clk.py)
    {90, 12, 1, 1, 3, 49, 0, 7},   // (probe builds only: NEGATIVE result) conv_wino4d_kernel [round 6]: 88's two blocks of a CU as the independent halves of ONE 12-wave workgroup (ai bit 5: LDS-counter barriers per half); filter kind 3
    {91, 12, 1, 1, 3, 49, 64, 7},  // 90 with s_memtime stamps (tools/wino4_clk.py)
    {92, 12, 1, 1, 3, 64, 0, 7},   // (probe builds only: NEGATIVE result) conv_wino4r_kernel [round 6]: F(4x4,3x3) on 16 x 32 regions with ROW-OWNER waves -- the row pass of the output transform in the accumulators, ONE exchange round per item (ai bit 6; conv_wino4r.hip); filter kind 3
    {93, 12, 1, 1, 3, 64, 64, 7},  // 92 with s_memtime stamps (tools/wino4_clk.py)
};
static const int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

extern "C" int egn_conv_num_configs(void) { return kNumConfigs; }
const ConvConfig* egn_conv_config(int cfg) {
  return (cfg >= 1 && cfg <= kNumConfigs) ? &kConfigs[cfg - 1] : nullptr;
}
extern "C" int egn_conv_config_info(int cfg, int* tile_m, int* tile_n) {
  if (cfg < 1 || cfg > kNumConfigs) return EGN_E_BADARG;
  if (tile_m) *tile_m = kConfigs[cfg - 1].tile_m();
  if (tile_n) *tile_n = kConfigs[cfg - 1].tile_n();
  return 0;
}

// 0 = direct kernels (wpack from egn_pack_conv_weight_f32), 1 = Winograd kernels (wpack from
// egn_wino_pack_weight_f32), -1 = not selectable (timing ablations, invalid ids)
// Families that were measured and lost (cfg 41 / 43: 4-wave and register-filter forms of conv_c48.hip; 45 / 46: the
// 4-wave Winograd kernel; 65: the first F(4x4,3x3) kernel; 67 / 68: two 4-wave blocks per CU) exist only in probe
// builds (-DEGN_PROBES: python -m egonet_amd.build --probes, used by tools/); the product library neither compiles
// nor launches them, and the timing-ablation / stamp builds (WRONG RESULTS) likewise.
static bool probe_only(const ConvConfig& c) {
  if (c.dma == 4) return c.id == 41 || c.id == 43;
  if (c.dma == 5) return (c.bi >> 4) != 0 || (c.bi & 15) <= 1 || (c.bi & 15) >= 10;
  // stamp builds; the half-block kernels (88 / 90) and the row-owner kernel (92): measured slower, profiles/r6_wino4h_*.txt, r6_wino4r_probe.txt
  if (c.dma == 7) return c.bi != 0 || (c.ai & (16 | 64)) != 0;
  return false;
}
extern "C" int egn_conv_config_kind(int cfg) {
  if (cfg < 1 || cfg > kNumConfigs) return -1;
  const ConvConfig& c = kConfigs[cfg - 1];
  if (c.dma == 3) return -1;  // retired ids
#ifndef EGN_PROBES
  if (probe_only(c)) return -1;
#endif
  if (c.dma == 7) return c.bi ? -1 : 3;   // F(4x4,3x3) filter in the register-feed layout (engine.pack_wino4_weight)
  if (c.dma != 5) return 0;
  if (c.bi >> 4) return -1;
  return (c.bi & 15) == 10 ? 2 : 1;     // 2: F(4x4,3x3) filter (engine.pack_wino43_weight)
}
// 1 = a probe build (ablation / stamp / retired configurations can be launched through egn_conv2d_f32(cfg))
extern "C" int egn_probe_build(void) {
#ifdef EGN_PROBES
  return 1;
#else
  return 0;
#endif
}

// kernel symbol of a config as rocprofv3 prints it (lets bench.py line its
// hipEvent timings up with the kernel-trace statistics)
extern "C" int egn_conv_config_name(int cfg, char* buf, int len) {
  if (cfg < 1 || cfg > kNumConfigs || !buf || len < 8) return EGN_E_BADARG;
  const ConvConfig& c = kConfigs[cfg - 1];
  if (c.dma == 6) { snprintf(buf, len, "conv_stem_kernel(ConvArgs)"); return 0; }
  if (c.dma == 7 && (c.ai & 3) == 2) { snprintf(buf, len, "void conv_wino4c_kernel<%d, %d>(ConvArgs)", c.bi, (c.ai & 4) ? 2 : 1); return 0; }
  if (c.dma == 7 && c.ai == 64) { snprintf(buf, len, "void conv_wino4r_kernel<%d>(ConvArgs)", c.bi); return 0; }
  if (c.dma == 7 && c.ai == 49) { snprintf(buf, len, "void conv_wino4d_kernel<%d>(ConvArgs)", c.bi); return 0; }
  if (c.dma == 7 && c.ai == 17) { snprintf(buf, len, "void conv_wino4h_kernel<%d>(ConvArgs)", c.bi); return 0; }
  if (c.dma == 7 && c.ai == 9) { snprintf(buf, len, "void conv_wino4w_kernel<%d>(ConvArgs)", c.bi); return 0; }
  if (c.dma == 7) { snprintf(buf, len, "void conv_wino4%s_kernel<%d>(ConvArgs)", c.ai == 5 ? "bk" : (c.ai ? "b" : ""), c.bi); return 0; }
  if (c.dma == 8) { snprintf(buf, len, "conv_fc_kernel(ConvArgs)"); return 0; }
  if (c.dma == 9) { snprintf(buf, len, "conv_s2r_kernel(ConvArgs)"); return 0; }
  if (c.dma == 5 && (c.bi & 15) == 10) { snprintf(buf, len, "void conv_wino43_kernel<0>(ConvArgs)"); return 0; }
  if (c.dma == 5 && ((c.bi & 15) == 11 || (c.bi & 15) == 12)) {
    snprintf(buf, len, "void conv_wino9_kernel<%s, 4, 3, 0, 2>(ConvArgs)", (c.bi & 15) == 11 ? "8, 16, 1" : "8, 8, 2");
    return 0;
  }
  if (c.dma == 5 && (c.bi & 15) >= 6) {
    static const char* geo9[4] = {"16, 16, 1, 8", "8, 8, 4, 8", "8, 8, 2, 4", "8, 16, 1, 4"};
    snprintf(buf, len, "void conv_wino9_kernel<%s, 3, 0, 4>(ConvArgs)", geo9[(c.bi & 15) - 6]);
    return 0;
  }
  if (c.dma == 5)
    // conv_wino8_kernel<TH, TW, TNB, ABL, NW, NT>: the symbol of the 48-channel co-tile build (NT = 3: the W48
    // widths); layers with Cout % 48 != 0 run the NT = 2 build of the same kernel
    if ((c.bi & 15) == 4) snprintf(buf, len, "void conv_wino8_kernel<8, 8, 2, 0, 4, 3>(ConvArgs)");
    else if ((c.bi & 15) == 5) snprintf(buf, len, "void conv_wino8_kernel<8, 16, 1, 0, 4, 3>(ConvArgs)");
    else if (c.bi & 2) snprintf(buf, len, "void conv_wino8_kernel<%s, %d, 8, 3>(ConvArgs)", (c.bi & 1) ? "8, 8, 4" : "16, 16, 1", c.bi >> 4);
    else snprintf(buf, len, "void conv_wino_kernel<%s, %d>(ConvArgs)", (c.bi & 1) ? "8, 8, 4" : "16, 16, 1", c.bi >> 4);
  else if (c.dma == 4 && c.bi == 2)
    snprintf(buf, len, "conv_c48t_kernel(ConvArgs)");
  else if (c.dma == 4 && c.bi == 1)
    snprintf(buf, len, "conv_c48r_kernel(ConvArgs)");
  else if (c.dma == 4)
    snprintf(buf, len, "void conv_c48_kernel<%d>(ConvArgs)", c.wm);
  else if (c.dma == 3)
    snprintf(buf, len, "(retired)");
  else if (c.dma)
    snprintf(buf, len, "void conv_dma_kernel<%d, %d, %d, %d, 8, 8, 0>(ConvArgs)", c.wm, c.wn, c.mt, c.nt);
  else
    snprintf(buf, len, "void conv_mfma_kernel<%d, %d, %d, %d, %d, %d>(ConvArgs)", c.wm, c.wn, c.mt, c.nt, c.ai,
             c.bi);
  return 0;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// LDS layout: [ K-loop stage buffers | epilogue sC (aliases them) ][ sPix: TM ints ]
static size_t lds_stage_bytes(const ConvArgs& a, const ConvConfig& cf) {
  size_t main_loop = (size_t)(EGN_CKQ * a.npixp + a.tps * EGN_CKQ * cf.tile_n()) * 16;
  if (cf.dma) main_loop *= 2;  // double-buffered stage
  // epilogue: 4 waves x (MT*16 rows) x (NT*16 + 4) floats
  const size_t epi = a.out_nchw ? 0 : (size_t)4 * cf.mt * 16 * (cf.nt * 16 + 4) * 4;
  return ((main_loop > epi ? main_loop : epi) + 15) & ~(size_t)15;
}
static size_t lds_bytes_for(const ConvArgs& a, const ConvConfig& cf) {
  if (cf.dma == 6) return egn_conv_stem_lds_bytes();
  if (cf.dma == 7) return egn_conv_wino4_lds_bytes(cf.ai);
  if (cf.dma == 8) return 0;
  if (cf.dma == 9) return egn_conv_s2r_lds_bytes();
  if (cf.dma == 5) return egn_conv_wino_lds_bytes(cf.bi, a.Cout);
  if (cf.dma == 4 && cf.bi == 2) return (size_t)(3 * 336 * EGN_CKQ + 3 * 9 * EGN_CKQ * 48) * 16;  // chunk ring + filter
  if (cf.dma == 4) return (size_t)(2 * 3 * 192 * EGN_CKQ + 3 * 9 * EGN_CKQ * 48) * 16;  // 2 halo buffers + filter
  return lds_stage_bytes(a, cf) + (size_t)cf.tile_m() * 4;
}

// Choose the spatial tile for a config: minimise (MFMA work incl. padding +
// LDS fill work) over power-of-two tile shapes, subject to the LDS budget and
// to the per-lane staging depth (ai / bi dwordx4 loads per stage).
static bool plan_tile(ConvArgs& a, const ConvConfig& cf, size_t lds_budget, double* cost_out) {
  if (cf.dma == 3) return false;  // retired ids
#ifndef EGN_PROBES
  if (probe_only(cf)) return false;   // never planned by the product library
#endif
  if (cf.dma == 6) {
    if (!egn_conv_stem_applies(a)) return false;
    a.TH = 16; a.TW = 16; a.TNB = 1; a.HH = 33; a.HW = 33;
    a.npix = 33 * 33; a.npixp = (a.npix + 15) & ~15; a.tps = 9;
    a.tiles_x = (a.Wo + 15) / 16;
    a.tiles_y = (a.Ho + 15) / 16;
    if (cost_out) *cost_out = 0.0;
    return true;
  }
  if (cf.dma == 8) {
    if (!egn_conv_fc_applies(a)) return false;
    a.TH = 1; a.TW = 1; a.TNB = 16; a.HH = 1; a.HW = 1;
    a.npix = 16; a.npixp = 16; a.tps = 1;
    a.tiles_x = 1;
    a.tiles_y = 1;
    if (cost_out) *cost_out = 0.0;
    return true;
  }
  if (cf.dma == 9) {
    // conv_s2r.hip: 2 x 8 output pixels per tile, 48-channel co-groups
    if (!egn_conv_s2r_applies(a)) return false;
    a.TH = 2; a.TW = 8; a.TNB = 1; a.HH = 5; a.HW = 17;
    a.npix = a.HH * a.HW; a.npixp = (a.npix + 15) & ~15; a.tps = 9;
    a.tiles_x = a.Wo / a.TW;
    a.tiles_y = a.Ho / a.TH;
    if (cost_out) *cost_out = 0.0;
    return true;
  }
  if (cf.dma == 7) {
    // conv_wino4.hip: 16 x 32 (ai = 0) / 16 x 16 (ai = 1) pixel regions of whole-region maps, or four whole 8 x 8
    // images per region (ai & 3 = 2; ai bit 2: K split), 48-channel co-tiles
    if (!egn_conv_wino4_applies(a, cf.ai)) return false;
    const int g7 = cf.ai & 3;
    a.TH = g7 == 2 ? 8 : 16; a.TW = g7 == 2 ? 8 : (g7 == 1 ? 16 : 32); a.TNB = g7 == 2 ? 4 : 1; a.HH = a.TH + 2; a.HW = a.TW + 2;
    a.npix = a.TNB * a.HH * a.HW; a.npixp = (a.npix + 15) & ~15; a.tps = 36;
    a.tiles_x = a.Wo / a.TW;
    a.tiles_y = a.Ho / a.TH;
    if (cost_out) *cost_out = 0.0;
    return true;
  }
  if (cf.dma == 5) {
    // conv_wino.hip: 3x3 stride-1 pad-1 NHWC layers with unpadded channel strides, even maps
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin % EGN_CK || a.cs_in != a.Cin ||
        a.cs_out != a.Cout || a.out_nchw || (a.Ho & 1) || (a.Wo & 1))
      return false;
    if ((cf.bi & 15) == 10) {
      // conv_wino43_kernel: 16 x 16 pixel tiles of whole-tile maps, 4-channel K steps, 48-channel co-tiles
      if (a.Cout % 48 || a.Cin % 4 || (a.Ho % 16) || (a.Wo % 16)) return false;
      a.TH = 16; a.TW = 16; a.TNB = 1; a.HH = 18; a.HW = 18;
      a.npix = 18 * 18; a.npixp = (a.npix + 15) & ~15; a.tps = 36;
      a.tiles_x = a.Wo / 16;
      a.tiles_y = a.Ho / 16;
      if (cost_out) *cost_out = 0.0;
      return true;
    }
    // co-tile 48 (4- and 8-wave kernels) or 32 (8-wave kernels only): egn_wino_cot in conv_wino.hip
    if (a.Cout % 48 != 0 && !(a.Cout % 32 == 0 && (cf.bi & 15) >= 2)) return false;
    const int vv = cf.bi & 15;      // 6..9: conv_wino9_kernel on 2..5's tiles; 11 / 12: its 8-channel-stage form on 5 / 4's
    const int geo = vv == 11 ? 5 : (vv == 12 ? 4 : (vv >= 6 ? vv - 4 : vv));
    if (geo == 4) { a.TH = 8; a.TW = 8; a.TNB = 2; }
    else if (geo == 5) { a.TH = 8; a.TW = 16; a.TNB = 1; }
    else if (geo == 1 || geo == 3) { a.TH = 8; a.TW = 8; a.TNB = 4; }
    else { a.TH = 16; a.TW = 16; a.TNB = 1; }
    if (a.TH == 8 && a.TW == 8 && (a.Ho > 8 || a.Wo > 8)) return false;   // the batched variant is for the 8 x 8 maps
    a.HH = a.TH + 2; a.HW = a.TW + 2;
    a.npix = a.TNB * a.HH * a.HW; a.npixp = (a.npix + 15) & ~15; a.tps = 16;
    a.tiles_x = (a.Wo + a.TW - 1) / a.TW;
    a.tiles_y = (a.Ho + a.TH - 1) / a.TH;
    if (cost_out) *cost_out = 0.0;
    return true;
  }
  if (cf.dma == 4) {
    // conv_c48.hip: exactly the 48 -> 48 3x3 stride-1 pad-1 NHWC layers, fixed 8 x 16 tile
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin != 48 || a.cs_in != 48 || a.Cout != 48 ||
        a.cs_out != 48 || a.out_nchw)
      return false;
    a.TH = cf.bi == 2 ? 16 : 8; a.TW = 16; a.TNB = 1; a.HH = a.TH + 2; a.HW = 18;
    a.npix = a.HH * 18; a.npixp = cf.bi == 2 ? 336 : 192; a.tps = 9;
    a.tiles_x = (a.Wo + 15) / 16;
    a.tiles_y = (a.Ho + a.TH - 1) / a.TH;
    if (cost_out) *cost_out = 0.0;
    return true;
  }
  const int tm = cf.tile_m();
  const int tn = cf.tile_n();
  double best = -1.0;
  ConvArgs bestA = a;
  for (int tw = 1; tw <= 64 && tw <= tm; tw *= 2) {
    if (tw < 4 && tw < a.Wo) continue;  // narrow tiles only for maps that narrow
    for (int th = 1; th * tw <= tm; th *= 2) {
      const int tnb = tm / (tw * th);
      if (tnb * tw * th != tm) continue;
      // no point in tiles much larger than the map
      if (tw >= 2 * a.Wo && tw > 1) continue;
      if (th >= 2 * a.Ho && th > 1) continue;
      ConvArgs c = a;
      c.TH = th; c.TW = tw; c.TNB = tnb;
      c.HH = (th - 1) * a.stride + a.KH;
      c.HW = (tw - 1) * a.stride + a.KW;
      c.npix = tnb * c.HH * c.HW;
      c.npixp = (c.npix + 15) & ~15;
      if (c.npix * EGN_CKQ > cf.ai * 256) continue;
      c.tiles_x = cdiv(a.Wo, tw);
      c.tiles_y = cdiv(a.Ho, th);
      const int tiles_b = cdiv(a.N, tnb);
      // taps per stage: as many as fit the LDS budget and the staging depth
      int tps = a.taps;
      c.tps = tps;
      while (tps > 1 && (lds_bytes_for(c, cf) > lds_budget || tps * EGN_CKQ * tn > cf.bi * 256)) {
        --tps;
        c.tps = tps;
      }
      if (lds_bytes_for(c, cf) > lds_budget || tps * EGN_CKQ * tn > cf.bi * 256) continue;
      // balance the stages (e.g. 9 taps -> 5+4 instead of 8+1)
      const int nst = cdiv(a.taps, tps);
      c.tps = cdiv(a.taps, nst);
      const double tiles = (double)c.tiles_x * c.tiles_y * tiles_b * cdiv(a.CoutP, tn);
      const double mfma = (double)tm * tn * a.taps * EGN_CK;  // per chunk per tile
      const double fill = (double)c.npix * EGN_CK * 24.0 + (double)a.taps * EGN_CK * tn * 12.0;
      // ties (1x1 convs have no halo): prefer contiguous pixels over many images
      const double cost = tiles * (mfma + fill + 4000.0 * nst + 64.0 * tnb + 8.0 * th);
      if (best < 0 || cost < best) { best = cost; bestA = c; }
    }
  }
  if (best < 0) return false;
  a = bestA;
  if (cost_out) *cost_out = best;
  return true;
}

static size_t budget_for(const ConvConfig& cf) {
  // staged: 3 blocks / CU; dma: 2 (ids 11..20) or 3 (ids 21..30) blocks / CU of the 160 KiB LDS
  return cf.dma == 1 ? 80 * 1024 : (cf.dma == 2 ? 53 * 1024 : 64 * 1024);
}

int egn_conv_plan(ConvArgs& a, int& cfg_id, size_t& lds_bytes) {
  if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.Cin <= 0 || a.Cout <= 0) return EGN_E_BADARG;
  if (a.cs_in % 4 || a.cs_in < a.Cin) return EGN_E_BADARG;
  if (!a.out_nchw && (a.cs_out % 4 || a.cs_out < a.Cout)) return EGN_E_BADARG;
  if (a.KH < 1 || a.KW < 1 || a.stride < 1 || a.pad < 0) return EGN_E_BADARG;
  a.Ho = (a.H + 2 * a.pad - a.KH) / a.stride + 1;
  a.Wo = (a.W + 2 * a.pad - a.KW) / a.stride + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return EGN_E_BADARG;
  // 32-bit byte offsets into x (buffer loads) and 32-bit pixel indices
  if ((double)a.N * a.H * a.W * a.cs_in * 4.0 >= 2147483648.0) return EGN_E_BADARG;
  if ((double)a.N * a.Ho * a.Wo >= 2147483648.0) return EGN_E_BADARG;
  a.CoutP = (a.Cout + 15) & ~15;
  a.nchunk = cdiv(a.Cin, EGN_CK);
  a.taps = a.KH * a.KW;
  if (cfg_id >= 1 && cfg_id <= kNumConfigs) {
    const ConvConfig& cf = kConfigs[cfg_id - 1];
    if (!plan_tile(a, cf, budget_for(cf), nullptr)) return EGN_E_LDS;
  } else {
    // cost model over the staged family (the tuner explores both families)
    double best = -1.0;
    int best_id = 0;
    ConvArgs bestA = a;
    for (int k = 0; k < kNumConfigs; ++k) {
      const ConvConfig& cf = kConfigs[k];
      if (cf.dma) continue;
      ConvArgs c = a;
      double cost;
      if (!plan_tile(c, cf, budget_for(cf), &cost)) continue;
      // mild preference for filling the chip: penalise grids below 256 blocks
      const double blocks = (double)c.tiles_x * c.tiles_y * cdiv(a.N, c.TNB) * cdiv(a.CoutP, cf.tile_n());
      if (blocks < 256.0) cost *= 256.0 / blocks > 4.0 ? 4.0 : 256.0 / blocks;
      if (best < 0 || cost < best) { best = cost; best_id = cf.id; bestA = c; }
    }
    if (best < 0) return EGN_E_LDS;
    a = bestA;
    cfg_id = best_id;
  }
  a.spix_off = (int)(lds_stage_bytes(a, kConfigs[cfg_id - 1]) / 16);
  lds_bytes = lds_bytes_for(a, kConfigs[cfg_id - 1]);
  // 32-bit byte offsets into y / res (buffer stores in the NHWC epilogue)
  if (!a.out_nchw && (double)a.N * a.Ho * a.Wo * a.cs_out * 4.0 >= 2147483648.0) return EGN_E_BADARG;
  return 0;
}

// rows of the partial-statistics table a launch with a.stats != NULL writes; 0 = config without fused
// BatchNorm statistics (a must be planned for cfg_id)
int egn_conv_stats_rows(const ConvArgs& a, int cfg_id) {
  if (cfg_id < 1 || cfg_id > kNumConfigs) return 0;
  const ConvConfig& cf = kConfigs[cfg_id - 1];
  if (cf.dma == 7) return cf.bi ? 0 : egn_conv_wino4_stats_rows(a, cf.ai);     // [round 5] conv_wino4s_kernel
  return cf.dma == 5 ? egn_conv_wino_stats_rows(a, cf.bi) : 0;
}

// ticket words (zeroed unsigned) a launch of cfg_id wants in a.tickets to run as ONE kernel; 0 = the config uses none
int egn_conv_ticket_count(const ConvArgs& a, int cfg_id) {
  if (cfg_id < 1 || cfg_id > kNumConfigs) return 0;
  const ConvConfig& cf = kConfigs[cfg_id - 1];
  return cf.dma == 7 ? egn_conv_wino4_tickets(a, cf.ai) : 0;
}

int egn_conv_launch(const ConvArgs& a, int cfg_id, hipStream_t stream) {
  if (cfg_id < 1 || cfg_id > kNumConfigs) return EGN_E_BADARG;
  const ConvConfig& cf = kConfigs[cfg_id - 1];
#ifndef EGN_PROBES
  if (egn_conv_config_kind(cfg_id) < 0) return EGN_E_BADARG;   // ablation / stamp / retired ids: probe builds only
#endif
  const size_t lds = lds_bytes_for(a, cf);
  if (cf.dma == 6) return egn_conv_launch_stem(a, lds, stream);
  if (cf.dma == 7) return egn_conv_launch_wino4(a, lds, cf.bi, cf.ai, stream);
  if (cf.dma == 8) return egn_conv_launch_fc(a, stream);
  if (cf.dma == 9) return egn_conv_launch_s2r(a, stream);
  if (cf.dma == 5) return egn_conv_launch_wino(a, lds, cf.bi, stream);
  if (cf.dma == 4) return egn_conv_launch_c48(a, lds, cf.bi == 2 ? -1 : cf.bi == 1 ? 0 : cf.wm, stream);
  if (cf.dma == 3) return EGN_E_BADARG;
  return cf.dma ? egn_conv_launch_dma(a, (cfg_id - 1) % 10 + 1, lds, stream)
                : egn_conv_launch_staged(a, cfg_id, lds, stream);
}
