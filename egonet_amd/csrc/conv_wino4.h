// conv_wino4.h -- device helpers shared by the F(4x4,3x3) kernel bodies: conv_wino4.hip (12 waves: 16 x 32 / 16 x 16 regions,
// four 8 x 8 images; one 48-channel co-tile per item) and conv_wino4w.hip (round 6: 96 output channels per item).
// Geometry constants of the bank-conflict-free halo order, the raw-ISA load / LDS wrappers whose waits the kernels
// place themselves, the 1-D transforms and the input transform of a wave's third (V = B^T d B).
// Reference: the 3x3 stride-1 convolutions of libs/model/heatmapModel/hrnet.py:49-76.
#pragma once
#include <stdlib.h>

#include "conv_common.h"
#include "wino4_pack.h"

typedef __attribute__((address_space(3))) void* lds_ptr_w4_t;

namespace {
constexpr int W4_NW = 12, W4_NTH = 64 * W4_NW;
constexpr int W4_CO = 48;
// Halo of the 16 x 32 region: 18 x 34 pixels x 8 channels, in 8-byte slots (one pixel, one channel PAIR) ordered
// [pair 0..3][x mod 4][y][x div 4 (pitch 10)]: the transform's lane (tile (ty, tx), channel kq of k-group g) reads
// pixel (4 ty + i, 4 tx + j) at slot (2 g + kq / 2) * 720 + 40 ty + tx + const(i, j), dword kq & 1.  ds_read_b32 is
// banked per 32-lane group on dword mod 32 (MI355X_MICROARCH.md): a group = 16 tiles x 2 channels of ONE pair plane,
// dword = 2 (40 ty + tx) + kq + const -> 32 different banks.  History (profiles/r3_wino4_ablations.txt,
// r3_pmc_sq_wino4.txt): pixel-major order put all 16 tiles of a group on one bank (the transform alone took 5 000
// cycles per stage); 16-byte quad planes left a 2-way conflict (SQ_LDS_BANK_CONFLICT 3.9 M cycles per launch).
// The loads fetch pixel-major (32 contiguous bytes per pixel) and each lane stores its two channel pairs.
template <int GEO>
struct W4G {
  static constexpr int NIMG = GEO == 2 ? 4 : 1;         // images of a region (GEO 2: whole 8 x 8 maps)
  static constexpr int RH = GEO == 2 ? 10 : 18, RW = GEO == 0 ? 34 : (GEO == 2 ? 10 : 18);     // halo pixels of an image
  // load-element row pitch: a 16-lane store group = 8 (GEO 0) / 4 ALIGNED pixels of one row
  static constexpr int RWP = GEO == 0 ? 40 : (GEO == 2 ? 12 : 20);
  static constexpr int XD = GEO == 0 ? 10 : (GEO == 2 ? 3 : 5);      // slots per row and plane (x div 4: 0..8 / 0..4 / 0..2)
  // plane / pair pitches carry a bank skew for the halo STORES (ds_write_b64 is served in groups of 16 lanes: 8 pixels x
  // 2 quads, GEO 1 / 2: 4 pixels x 4 quads; without the skew the quads of a pixel fell on one bank: 2- / 4-way conflicts).
  // GEO 0: pixel x -> 8 (x & 3) + 2 (x >> 2) dwords, quad -> + 4: 32 banks.  GEO 1 / 2: pixel -> 2 (x & 3), quad -> 8 q.
  // GEO 3 (conv_wino4h.hip: GEO 1's region with GEO 0's 8-channel stages -- 8 pixels x 2 quads per store group): GEO 0's rule.
  // The transform's READS are per (i, j) and per pair plane: they only see XD and the image bases (bank-free as before).
  static constexpr int PLANE = GEO == 0 ? RH * XD : (GEO == 1 ? 97 : (GEO == 2 ? 145 : 100));     // 2 PLANE mod 32 = 8 / 2 / 2 / 8 dwords
  static constexpr int PAIR = GEO == 0 || GEO == 3 ? 4 * PLANE + 1 : (GEO == 1 ? 394 : 586);   // 8-byte slots per channel pair; 4 PAIR mod 32 = 4 / 8 / 8 / 4 dwords
  static constexpr int QPP = GEO == 1 || GEO == 2 ? 4 : 2;           // channel quads per pixel and stage (16 / 8 channels)
  static constexpr int NP = GEO == 2 ? 3 : 2;           // halo load pieces per wave and stage
  static constexpr int HSLOT = 2 * QPP * PAIR;          // 2884 / 3152 / 4688 / 1604 slots
  static constexpr int HBYTES = (GEO == 2 ? 38 : (GEO == 3 ? 13 : 26)) * 1024;   // halo slots + 1 KB parking for the idle load lanes (GEO 3: none)
  static constexpr int SBYTES = 16 * QPP;               // bytes of a pixel's channels of one stage
  static constexpr int NMT = GEO == 0 ? 2 : 1;          // m-tiles of a region
  static constexpr int NKK = GEO == 1 || GEO == 2 ? 2 : 1;           // k-groups multiplied per filter wait ("k-group pair")
  static constexpr int TWX = GEO == 0 ? 8 : (GEO == 2 ? 2 : 4);      // tiles per image row
  static constexpr int RGW = GEO == 0 ? 32 : (GEO == 2 ? 8 : 16);    // region width in pixels
  static constexpr int RGH = GEO == 2 ? 8 : 16;                      // region height
  static constexpr int VPT = GEO == 3 ? 512 : 1024;     // bytes of a frequency point in a V buffer ([k-groups of a stage][64 lanes] floats)
  // slot of image `img` inside a plane (GEO 2): 30 slots per image + a skew of 4 per image and 4 more per image PAIR, so
  // that the 16 tiles of a read group (img 0..3 x ty 0..1 (12 slots) x tx 0..1) fall on 16 different even dwords mod 32
  __host__ __device__ static constexpr int imgbase(int img) { return GEO == 2 ? 34 * img + 4 * (img >> 1) : 0; }
  // halo slot of lane tile `li` (pixel (0, 0) of the tile, x & 3 == 0 plane)
  __host__ __device__ static constexpr int tileslot(int li, int tw) {
    return GEO == 0 ? 4 * XD * (2 * (tw >> 1) + (li >> 3)) + (li & 7)
                    : (GEO == 2 ? imgbase(li >> 2) + 4 * XD * ((li >> 1) & 1) + (li & 1) : 4 * XD * (li >> 2) + (li & 3));
  }
};
// GEO 1 (tests/test_wino4_design_cpu.py): slot (4 XD ty + tx) -> dword 40 ty + 2 tx + (kq & 1): ty 0..3 -> banks
// +0, +8, +16, +24 -- 32 different banks per 32-lane group again.  GEO 2: slot imgbase(img) + 12 ty + tx -> dwords
// {0, 2, 24, 26} + {0, 68, 144, 212} mod 32 = {0, 4, 16, 20}: 16 different even banks.
constexpr int W4_VBYTES = 36 * 1024;                  // [pt][mt][g][lane] floats (GEO 1 / 2: [pt][g 0..3][lane])
constexpr int W4_V0 = 0, W4_V1 = W4_VBYTES, W4_H0 = 2 * W4_VBYTES;
template <int GEO>
constexpr int w4_lds_bytes() { return W4_H0 + 2 * W4G<GEO>::HBYTES; }      // 126 976 B (GEO 0 / 1), 151 552 B (GEO 2)
constexpr int W4_XBYTES = 36 * 3 * 1024;              // exchange [pt][nt][lane] float4: 110 592 B
static_assert(W4_XBYTES <= w4_lds_bytes<0>(), "the exchange reuses the stage buffers");
static_assert(W4G<0>::HSLOT * 8 + 1024 <= W4G<0>::HBYTES && W4G<1>::HSLOT * 8 + 1024 <= W4G<1>::HBYTES &&
                  W4G<2>::HSLOT * 8 + 1024 <= W4G<2>::HBYTES,
              "the halo and the parking slots of the idle load lanes fit the buffer");
static_assert(W4G<0>::QPP * W4G<0>::RH * W4G<0>::RWP <= W4G<0>::NP * W4_NW * 64 &&
                  W4G<1>::QPP * W4G<1>::RH * W4G<1>::RWP <= W4G<1>::NP * W4_NW * 64 &&
                  W4G<2>::QPP * W4G<2>::NIMG * W4G<2>::RH * W4G<2>::RWP <= W4G<2>::NP * W4_NW * 64,
              "the load pieces of the waves cover the halo");
static_assert(W4G<2>::imgbase(3) + W4G<2>::RH * W4G<2>::XD <= W4G<2>::PLANE && w4_lds_bytes<2>() + 12 * 96 * 8 <= 160 * 1024 - 512,
              "GEO 2: four images fit a plane, the buffers fit the CU");
constexpr int W4_UKG = W4_NW * 3 * 64 * 4;            // filter floats of one (co-tile, stage, k-group): 9216 (9 of 12 used)
static_assert(W4_UKG == W4P_UKG && W4_CO == W4P_CO, "wino4_pack.h packs this kernel's layout");
constexpr unsigned W4_PAST = 0x80000000u;             // scalar byte offset past every buffer (tensors stay below 2 GB)
}  // namespace

// one LDS dword at a VGPR byte address + immediate (see conv_wgrad_wino.hip: the compiler's ds_read2 pairing
// costs a v_add per pair); the values are tied to w4_landed's s_waitcnt before use
template <int OFF>
__device__ __forceinline__ float w4_lds(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b32 immediate");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void w4_landed6(float (&a)[6], float (&b)[6]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(b[0]), "+v"(b[1]),
                 "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]));
}
__device__ __forceinline__ void w4_tie6(float (&a)[6]) {
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
}
// filter loads: dwordx4 (a burst of 20 single-dword loads per wave and stage cost every wave ~1 000 cycles of
// issue time behind the barrier, profiles/r3_wino4_timeline_v1.txt)
template <int OFF>
__device__ __forceinline__ f32x4 w4_gld4(u32x4 rsrc, unsigned voff, unsigned soff) {
  static_assert(OFF >= 0 && OFF < 4096, "the immediate offset of a buffer instruction has 12 bits (the assembler truncates silently)");
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void w4_vm_landed3(f32x4 (&b)[3]) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void w4_vm_landed2(f32x4 (&h)[2]) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(h[0]), "+v"(h[1]) : "n"(N));
}
__device__ __forceinline__ void w4_vm_landedH(f32x4 (&h)[2]) { w4_vm_landed2<0>(h); }
__device__ __forceinline__ void w4_vm_landedH(f32x4 (&h)[3]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]));
}
// the filter registers of a wait group (one k-group: 3 dwordx4; GEO 1's k-group pair: 6), tied to the s_waitcnt
__device__ __forceinline__ void w4_vm_landedB(f32x4 (&b)[1][3]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]));
}
__device__ __forceinline__ void w4_vm_landedB(f32x4 (&b)[2][3]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]));
}
template <int OFF>
__device__ __forceinline__ void w4_xwr(unsigned addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void w4_xwr2(unsigned addr, float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int PT>
__device__ __forceinline__ float w4_xrd(unsigned xr0, unsigned xr1) {
  if constexpr (PT < 18) return w4_lds<PT * 3072>(xr0);
  else return w4_lds<(PT - 18) * 3072>(xr1);
}
__device__ __forceinline__ unsigned w4_udiv(unsigned x, unsigned magic) { return magic ? __umulhi(x, magic) : x; }

// Item order: what the XCD of a block (block index mod 8 -- blocks w, w + 8, ... stay on one XCD; a matter of speed only)
// owns.  The transformed filter is 36 x 4 bytes per (ci, co): 21 MB at 384 -> 384 channels, 5.3 MB at 192 -> 192 -- more
// than an XCD's 4 MB of L2, and a layer's filter arrives cold from HBM.
//   0  regions: 8 consecutive regions on the 8 XCDs, every XCD streams the WHOLE filter (small filters: 48 / 96 channels);
//   1  nct in {2, 4} (from W4_COX_MIN_NCT up): co-tile = xcd % nct, the XCD takes every (8 / nct)-th (region, half);
//   2  nct a multiple of 8: co-tile = 8 j + xcd -- each XCD reads its share of the filter once and keeps it, and reads
//      the halos of all regions instead (6 MB per 64 images at 384 channels).
// Measured on the 64-crop network (profiles/r4_wino4c.txt): 384 -> 384 @ 8 x 8 with regions on the XCDs 13.60 ms per
// batch, with co-tiles on them 13.17 ms.
#ifndef W4_COX_MIN_NCT
#define W4_COX_MIN_NCT 4
#endif
__host__ __device__ inline int w4_item_mode(int nct) {
#ifdef W4_NO_COTILE_XCD
  return 0;
#else
  if ((nct & 7) == 0) return 2;
  return nct >= W4_COX_MIN_NCT && (nct == 2 || nct == 4) ? 1 : 0;
#endif
}
__host__ __device__ inline int w4_item_count(int mode, int nreg, int nct, int ks) {
  if (mode == 2) return nreg * nct * ks;
  if (mode == 1) return 8 * ((nreg * ks + 8 / nct - 1) / (8 / nct));
  return ((nreg + 7) >> 3) * nct * ks * 8;
}

// 1-D input transform of six values (12 instructions)
__device__ __forceinline__ void w4_bt(const float (&t)[6], float (&o)[6]) {
  o[0] = __builtin_fmaf(4.f, t[0], __builtin_fmaf(-5.f, t[2], t[4]));
  const float u = __builtin_fmaf(-4.f, t[2], t[4]), v = __builtin_fmaf(-4.f, t[1], t[3]);
  o[1] = u + v;
  o[2] = u - v;
  const float p = t[4] - t[2], q = t[3] - t[1];
  o[3] = __builtin_fmaf(2.f, q, p);
  o[4] = __builtin_fmaf(-2.f, q, p);
  o[5] = __builtin_fmaf(4.f, t[1], __builtin_fmaf(-5.f, t[3], t[5]));
}
// 1-D output transform of six values (10 instructions)
__device__ __forceinline__ void w4_at(const float (&m)[6], float (&y)[4]) {
  const float p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
  y[0] = m[0] + p + r;
  y[1] = __builtin_fmaf(2.f, s, q);
  y[2] = __builtin_fmaf(4.f, r, p);
  y[3] = __builtin_fmaf(8.f, s, q) + m[5];
}

// halo of stage parity P -> V, for this wave's (m-tile, k-group) share and its THIRD of the frequency rows:
// PART 0 rows {0, 5}, 1 rows {1, 2}, 2 rows {3, 4} -- 48 VALU instructions, 36 / 24 / 24 LDS reads, 12 writes each.
// hb0: per-lane byte base in halo buffer 0 (the immediates reach both buffers); vw0: base in THIS parity's V buffer.
template <int P, int PART, int GEO>
__device__ __forceinline__ void w4_transform(unsigned hb0, unsigned vw0) {
  typedef W4G<GEO> Q;
  static_assert(Q::HBYTES + (3 * Q::PLANE + 5 * Q::XD + 1) * 8 < 65536, "halo immediates");
  constexpr int HO = P ? Q::HBYTES : 0;
#define W4_D(I, J) w4_lds<HO + (((J) & 3) * Q::PLANE + (I)*Q::XD + ((J) >> 2)) * 8>(hb0)
#define W4_WR(PT, VAL) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(vw0), "v"(VAL), "n"((PT)*Q::VPT) : "memory")
#define W4_ROW(FI, O)                                                                                   \
  W4_WR((FI)*6 + 0, (O)[0]); W4_WR((FI)*6 + 1, (O)[1]); W4_WR((FI)*6 + 2, (O)[2]);                      \
  W4_WR((FI)*6 + 3, (O)[3]); W4_WR((FI)*6 + 4, (O)[4]); W4_WR((FI)*6 + 5, (O)[5]);
#define W4_RD6(DST, I)                                                                                  \
  DST[0] = W4_D(I, 0); DST[1] = W4_D(I, 1); DST[2] = W4_D(I, 2); DST[3] = W4_D(I, 3); DST[4] = W4_D(I, 4); DST[5] = W4_D(I, 5);
  if constexpr (PART == 0) {
    // row 0: T = 4 d0 - 5 d2 + d4; row 5: T = 4 d1 - 5 d3 + d5
    {
      float x[6], y[6], z[6], t[6], o[6];
      W4_RD6(x, 0) W4_RD6(y, 2) W4_RD6(z, 4)
      w4_landed6(x, y);
      w4_tie6(z);
#pragma unroll
      for (int j = 0; j < 6; ++j) t[j] = __builtin_fmaf(4.f, x[j], __builtin_fmaf(-5.f, y[j], z[j]));
      w4_bt(t, o);
      W4_ROW(0, o)
    }
    {
      float x[6], y[6], z[6], t[6], o[6];
      W4_RD6(x, 1) W4_RD6(y, 3) W4_RD6(z, 5)
      w4_landed6(x, y);
      w4_tie6(z);
#pragma unroll
      for (int j = 0; j < 6; ++j) t[j] = __builtin_fmaf(4.f, x[j], __builtin_fmaf(-5.f, y[j], z[j]));
      w4_bt(t, o);
      W4_ROW(5, o)
    }
  } else {
    float d1[6], d2[6], d3[6], d4[6], ta[6], tb[6], o[6];
    W4_RD6(d1, 1) W4_RD6(d2, 2) W4_RD6(d3, 3) W4_RD6(d4, 4)
    w4_landed6(d1, d2);
    w4_tie6(d3);
    w4_tie6(d4);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if constexpr (PART == 1) {        // rows 1, 2: u = d4 - 4 d2, v = d3 - 4 d1
        const float u = __builtin_fmaf(-4.f, d2[j], d4[j]), v = __builtin_fmaf(-4.f, d1[j], d3[j]);
        ta[j] = u + v;
        tb[j] = u - v;
      } else {                          // rows 3, 4: p = d4 - d2, q = d3 - d1
        const float p = d4[j] - d2[j], q = d3[j] - d1[j];
        ta[j] = __builtin_fmaf(2.f, q, p);
        tb[j] = __builtin_fmaf(-2.f, q, p);
      }
    }
    w4_bt(ta, o);
    if constexpr (PART == 1) { W4_ROW(1, o) } else { W4_ROW(3, o) }
    w4_bt(tb, o);
    if constexpr (PART == 1) { W4_ROW(2, o) } else { W4_ROW(4, o) }
  }
#undef W4_RD6
#undef W4_D
#undef W4_WR
#undef W4_ROW
}
