// conv_wino.hip -- 3x3 / stride 1 / pad 1 convolution (NHWC, fp32) as FUSED WINOGRAD F(2x2, 3x3)
// on the fp32 matrix pipe.
//
// Why: HRNet's 3x3 s1 layers (hrnet.py:68-92 BasicBlock convs, 208 of the 306 conv launches, 89 % of
// the forward's FLOPs) are bound by the fp32 FMA rate, not by HBM (AI 72..600 FLOP/B).  Against an
// FMA roof the only lever left after tuning the direct kernels (conv_c48.hip: 74.5 % of peak) is to
// issue fewer FMAs: Winograd's minimal filtering computes a 2x2 output patch from a 4x4 input patch
// with 16 multiplies per (ci, co) instead of 36 -- 2.25x fewer MACs -- at the price of cheap +/-
// transforms.  Unfused (transform kernels + batched GEMM) it would move 4x the activation bytes
// through HBM and lose; here BOTH data transforms are lane-local register arithmetic around the same
// LDS-halo / MFMA structure the direct kernels use, so HBM sees exactly the direct kernel's traffic:
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray, F(2x2,3x3))
//
//   * GEMM view per frequency f = (i,j) of the 4x4 transform domain:
//         M_f[tile][co] = sum_ci V_f[tile][ci] * U_f[ci][co]
//     M = 16 Winograd tiles (2x2 output pixels each) per wave, N = 48 output channels, K = Cin in
//     chunks of 16.  v_mfma_f32_16x16x4_f32, the fragment mapping of the direct kernels: lane
//     (li = l&15, kq = l>>4) supplies A[tile li][channels 4kq..4kq+3] and B[4kq..][co li], owns
//     C[tiles 4kq+r][co li].
//   * U = G g G^T is precomputed (host, float64 -> fp32) and packed like a 4x4-tap filter:
//     [co-tile][chunk][f][quad][48][4] floats, so one (chunk, co-tile) slab is 48 KB contiguous and
//     a lane's B fragment of (f, nt) is one ds_read_b128.
//   * input transform in registers: a lane reads the 4x4 patch of ITS tile and ITS channel quad from
//     the LDS halo (16 ds_read_b128), forms V = B^T d B with 128 adds and uses the 16 results
//     directly as the A operands of the 16 frequencies.  Nothing transformed ever touches LDS.
//   * all 16 frequencies of a (tile, co) accumulate in the SAME lane (192 accumulator registers per
//     wave, one wave per SIMD owns the 512-entry file), so the output transform A^T M A is lane
//     local too: 24 adds per 2x2 patch, then scale/shift (+residual) + activation and dword stores
//     (16 lanes cover 64 contiguous bytes of a pixel, as in conv_c48.hip).
//   * persistent blocks walk over (spatial tile, co-tile) items; a K step = one 16-channel chunk:
//     halo chunk + U slab arrive by LDS-DMA in double-buffered stages, ONE barrier per step.
//
// Two kernels in this file.  conv_wino_kernel (configs 45 / 46) is the design just described: 4 waves,
// all 16 frequencies in one wave.  conv_wino8_kernel (configs 51 / 52 / 56 / 57, further down) is what
// the measured table selects: the frequency rows of an m-tile split between two waves (96 accumulators,
// two waves per SIMD), a partial exchange through LDS at the end of an item, BatchNorm statistics in
// the epilogue for the training step.  Kernel symbol <-> config: egn_conv_config_name (conv_plan.hip).
//
// Numerics: exact fp32 products, fp32 accumulation; the +/- transforms add a few ulp relative to
// the direct sum (the filter transform is done in float64).  Not bit-identical to the direct
// kernels -- the parity bar is the reference's (1e-3 px on key-points, arg-max indices exact).
#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_wino_t;

// Output channels per block (co-tile, egn_wino_cot in egn_internal.h): 48 (NT = 3 MFMA column tiles)
// where Cout allows -- the widths of HRNet-W48 (48 / 96 / 192 / 384) -- else 32 (NT = 2): the W32 /
// Pedestrian widths (32 / 64 / 128 / 256) and the 64-channel stem layers.  The SAME rule in the filter
// packing, the planner and the launcher.

namespace {
constexpr int WN_CO = 48;                          // the 4-wave kernel: 48 output channels per block (NT = 3)
constexpr int WN_NT = 3;
constexpr int WN_USLOTS = 16 * EGN_CKQ * WN_CO;    // float4 per (chunk, co-tile) slab of U: 3072 = 48 KB
constexpr int WN_NTH = 256;
constexpr int WN_UIT = WN_USLOTS / WN_NTH;         // 12 DMA instructions per lane and slab
}  // namespace

// Geometry of a block tile (64 Winograd tiles = 4 waves x 16) and the LDS image of its halo chunk.
//
// The halo lives in LDS as FOUR CHANNEL-QUAD PLANES  sH[quad][pixel slot]  (16 B per slot), not pixel
// major: a patch read (one ds_read_b128 per lane) is served in 4 groups of 16 lanes, each group holding
// all 16 tiles (li) of the wave with only two quad values, so it is conflict free exactly when the 16
// tiles' pixel slots are distinct mod 16 (and the plane size is a multiple of 256 B).  Tiles step by TWO
// pixels, so plain row-major slots collide; each geometry therefore skews its rows / images:
//   <16,16,1>  tile li = (ty = li>>3, tx = li&7): slot = y*24 + ((y>>1)&1) + x.  The two tile rows of a
//              wave are 2 pixel rows apart = 48 slots (= 0 mod 16) +- 1 from the skew: 2tx + {0,1}.
//   <8,8,4>    wave = tile row, li = (image b = li>>2, tx = li&3): slot = b*112 + g(b) + y*10 + x with
//              g = {0,1,8,9}: 2tx + g(b) covers 0..15.
// (pixel-major slots measured 4-way conflicts: 16 instead of 4 LDS cycles per read, 16 % of a K step.)
template <int TH, int TW, int TNB>
struct WinoGeom;

template <>
struct WinoGeom<16, 16, 1> {
  static constexpr int TH = 16, TW = 16, TNB = 1, HH = 18, HW = 18;
  static constexpr int RP = 24, PLANE = 432;     // 18 rows x 24 slots; 6912 B = 27 x 256
  static constexpr int ROFF = RP;                // slot step of one patch row
  // DMA slot p of a plane -> halo pixel (b, hy, hx) or -1
  static __device__ __forceinline__ int decode(int p) {
    const int y = p / RP, x = p - y * RP - ((y >> 1) & 1);
    return (y < HH && x >= 0 && x < HW) ? ((y << 8) | x) : -1;
  }
  // slot of the patch origin of tile li of wave w, for patch rows {0,1} (k = 0) / {2,3} (k = 1)
  static __device__ __forceinline__ int patch_base(int w, int li, int k) {
    const int tyl = li >> 3, tx = li & 7;
    return (4 * w + 2 * tyl) * RP + 2 * tx + ((tyl + k) & 1);
  }
  // tile index (b << 16 | ty << 8 | tx) of MFMA result row m (0..15) of wave w
  static __device__ __forceinline__ int out_tile(int w, int m) { return ((2 * w + (m >> 3)) << 8) | (m & 7); }
};

template <>
struct WinoGeom<8, 8, 4> {
  static constexpr int TH = 8, TW = 8, TNB = 4, HH = 10, HW = 10;
  static constexpr int RP = 10, IMGP = 112, PLANE = 448;   // 7168 B = 28 x 256
  static constexpr int ROFF = RP;
  static __device__ __forceinline__ int skew(int b) { return (b & 1) + ((b & 2) << 2); }  // 0, 1, 8, 9
  static __device__ __forceinline__ int decode(int p) {
    const int b = p / IMGP, r = p - b * IMGP - skew(b);
    const int y = r / RP, x = r - y * RP;
    return (r >= 0 && r < HH * RP) ? ((b << 16) | (y << 8) | x) : -1;
  }
  static __device__ __forceinline__ int patch_base(int w, int li, int) {
    const int b = li >> 2, tx = li & 3;
    return b * IMGP + skew(b) + 2 * w * RP + 2 * tx;
  }
  static __device__ __forceinline__ int out_tile(int w, int m) { return ((m >> 2) << 16) | (w << 8) | (m & 3); }
};

// half of the 16 x 16 tile (8 rows x 16 columns = 32 Winograd tiles, conv_wino8_kernel with 4 waves): twice
// the work items for maps / batches that leave CUs idle with the 64-tile form.  Same slots, fewer rows.
template <>
struct WinoGeom<8, 16, 1> {
  static constexpr int TH = 8, TW = 16, TNB = 1, HH = 10, HW = 18;
  static constexpr int RP = 24, PLANE = 256;     // 10 rows x 24 slots = 240, padded to 16 x 256 B
  static constexpr int ROFF = RP;
  static __device__ __forceinline__ int decode(int p) {
    const int y = p / RP, x = p - y * RP - ((y >> 1) & 1);
    return (y < HH && x >= 0 && x < HW) ? ((y << 8) | x) : -1;
  }
  static __device__ __forceinline__ int patch_base(int w, int li, int k) {
    const int tyl = li >> 3, tx = li & 7;
    return (4 * w + 2 * tyl) * RP + 2 * tx + ((tyl + k) & 1);
  }
  static __device__ __forceinline__ int out_tile(int w, int m) { return ((2 * w + (m >> 3)) << 8) | (m & 7); }
};

// two 8 x 8 images per block (32 Winograd tiles = 2 m-tiles; conv_wino8_kernel with 4 waves): m-tile mt
// holds tile rows {2mt, 2mt+1} of both images, li = (b = li>>3, row = (li>>2)&1, tx = li&3):
// slot = b*128 + b + y*12 + x -- 2tx + {0, 24 = 8 mod 16} + {0, 1} covers 0..15.
template <>
struct WinoGeom<8, 8, 2> {
  static constexpr int TH = 8, TW = 8, TNB = 2, HH = 10, HW = 10;
  static constexpr int RP = 12, IMGP = 128, PLANE = 256;   // 4096 B = 16 x 256
  static constexpr int ROFF = RP;
  static __device__ __forceinline__ int decode(int p) {
    const int b = p / IMGP, r = p - b * IMGP - b;
    const int y = r / RP, x = r - y * RP;
    return (r >= 0 && r < HH * RP && x < HW) ? ((b << 16) | (y << 8) | x) : -1;
  }
  static __device__ __forceinline__ int patch_base(int mt, int li, int) {
    const int b = li >> 3, tyl = (li >> 2) & 1, tx = li & 3;
    return b * IMGP + b + 2 * (2 * mt + tyl) * RP + 2 * tx;
  }
  static __device__ __forceinline__ int out_tile(int mt, int m) {
    return ((m >> 3) << 16) | ((2 * mt + ((m >> 2) & 1)) << 8) | (m & 3);
  }
};

template <int TH, int TW, int TNB>
struct WinoDims {
  using G = WinoGeom<TH, TW, TNB>;
  static constexpr int SLOTS = EGN_CKQ * G::PLANE;
  static constexpr int IT = (SLOTS + WN_NTH - 1) / WN_NTH;  // DMA instructions per lane and halo chunk
  static constexpr int BUF = IT * WN_NTH;                   // every wave issues IT whole instructions
  static_assert(G::PLANE % 16 == 0, "planes must start on a 256-byte boundary");
  static_assert(TNB * (TH / 2) * (TW / 2) == 64 || TNB * (TH / 2) * (TW / 2) == 32, "m-tiles of 16 Winograd tiles");
};

__device__ __forceinline__ void wino_dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "m0");
}
__device__ __forceinline__ unsigned wino_lds_addr(const void* p) {
  return (unsigned)(__UINTPTR_TYPE__)(lds_ptr_wino_t) const_cast<void*>(p);
}
__device__ __forceinline__ float wino_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void wino_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// ABL != 0 are timing ablations (wrong results by construction; cfg ids 47..50, tools/wino_probe.py only):
//   bit 0 no DMA after the prologue, bit 1 no patch reads / input transform, bit 2 no residual loads /
//   output stores, bit 3 no barrier / waitcnt at the step tops
template <int TH, int TW, int TNB, int ABL = 0>
__global__ __launch_bounds__(WN_NTH, 1) void conv_wino_kernel(ConvArgs a) {
  using G = WinoGeom<TH, TW, TNB>;
  using D = WinoDims<TH, TW, TNB>;
  constexpr int IT = D::IT;
  constexpr int NPIECE = IT + WN_UIT;  // DMA instructions per lane and K step
  extern __shared__ float4 smem[];
  float4* sU = smem;                  // [2][WN_USLOTS]   (f, quad, co) of one chunk and co-tile
  float4* sH = smem + 2 * WN_USLOTS;  // [2][D::BUF]      quad planes of one halo chunk

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;
  const int kq = lane >> 4;

  const int C = a.Cin;        // cs_in == Cin (planner)
  const int Co = a.Cout;      // cs_out == Cout, multiple of 48
  const int nct = Co / WN_CO;
  const int nchunk = a.nchunk;

  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const unsigned long long uaddr = reinterpret_cast<unsigned long long>(a.w);
  const u32x4 rxv = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu,
                     (unsigned)((size_t)a.N * a.H * a.W * C * 4), 0x00020000u};
  const u32x4 ruv = {(unsigned)uaddr, (unsigned)(uaddr >> 32) & 0xffffu,
                     (unsigned)((size_t)nct * nchunk * WN_USLOTS * 16), 0x00020000u};
  const unsigned out_bytes = (unsigned)((size_t)a.N * a.Ho * a.Wo * Co * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, out_bytes, 0x00020000);

  // halo slot e = it*256 + tid -> (quad plane, pixel slot) -> (image b, hy, hx); item independent
  int hmeta[IT];  // b << 16 | hy << 8 | hx, -1 = pad slot;  the quad is e / PLANE
  unsigned hq[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * WN_NTH + tid;
    const int q = e / G::PLANE;
    hmeta[it] = q < EGN_CKQ ? G::decode(e - q * G::PLANE) : -1;
    hq[it] = (unsigned)q * 16u;
  }

  // this lane's patch origin (float4 index in a halo buffer) for patch rows {0,1} and {2,3}
  const int pb01 = kq * G::PLANE + G::patch_base(wave, li, 0);
  const int pb23 = kq * G::PLANE + G::patch_base(wave, li, 1);
  // the 4 tiles whose results it owns (rows 4kq + r of the MFMA tile): b << 16 | ty << 8 | tx
  int og[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) og[r] = G::out_tile(wave, 4 * kq + r);

  const int tiles_xy = a.tiles_x * a.tiles_y;
  const int ntile = tiles_xy * ((a.N + TNB - 1) / TNB);
  const int nwork = ((ntile + 7) >> 3) * nct * 8;  // work index space, see WINO_ITEM

// work index w -> (spatial tile, co-tile).  Blocks w, w+8, w+16 ... run on one XCD (block b -> XCD
// b % 8): the co-tiles of ONE spatial tile are consecutive slots of one XCD, so the halo they all read
// is fetched into that XCD's L2 once.
#define WINO_ITEM(Wi, TILE_, CT_)            \
  {                                          \
    const int x_ = (Wi)&7, q_ = (Wi) >> 3;   \
    TILE_ = (q_ / nct) * 8 + x_;             \
    CT_ = q_ - (q_ / nct) * nct;             \
  }
// halo DMA byte offsets (chunk 0) of a spatial tile; out-of-image and pad slots get the OOB offset
// and are written as zeros by the same DMA (= the convolution's zero padding)
#define WINO_DOFF(TILE_, OUT)                                                                       \
  {                                                                                                 \
    const int tb_ = (TILE_) / tiles_xy;                                                             \
    const int r_ = (TILE_)-tb_ * tiles_xy;                                                          \
    const int ty_ = r_ / a.tiles_x, tx_ = r_ - ty_ * a.tiles_x;                                     \
    const int n0_ = tb_ * TNB, iy0_ = ty_ * TH - 1, ix0_ = tx_ * TW - 1;                            \
    _Pragma("unroll") for (int it = 0; it < IT; ++it) {                                             \
      const int m_ = hmeta[it];                                                                     \
      const int n_ = n0_ + (m_ >> 16), iy_ = iy0_ + ((m_ >> 8) & 255), ix_ = ix0_ + (m_ & 255);     \
      const bool in_ = m_ >= 0 && (TILE_) < ntile && n_ < a.N && iy_ >= 0 && iy_ < a.H && ix_ >= 0 && ix_ < a.W; \
      OUT[it] = in_ ? (unsigned)(((n_ * a.H + iy_) * a.W + ix_) * C) * 4u + hq[it] : EGN_OOB;       \
    }                                                                                               \
  }
// DMA instruction K (0 .. NPIECE-1) of a K step: halo chunk CH of the tile with offsets OFF (K < IT),
// then the U slab (CT, CH), into stage buffer P
#define WINO_PIECE(K, P, OFF, CT, CH)                                                                \
  {                                                                                                  \
    if ((K) < IT) {                                                                                  \
      wino_dma16(rxv, wino_lds_addr(sH + (P)*D::BUF + wave * 64) + (K)*WN_NTH * 16, OFF[(K) < IT ? (K) : 0], \
                 (unsigned)(CH)*64u);                                                                \
    } else {                                                                                         \
      wino_dma16(ruv, wino_lds_addr(sU + (P)*WN_USLOTS + wave * 64) + ((K)-IT) * WN_NTH * 16, (unsigned)tid * 16u, \
                 (unsigned)(((CT)*nchunk + (CH)) * WN_USLOTS) * 16u + ((K)-IT) * WN_NTH * 16);       \
    }                                                                                                \
  }
#define WINO_ISSUE(P, OFF, CT, CH) \
  { _Pragma("unroll") for (int k_ = 0; k_ < NPIECE; ++k_) WINO_PIECE(k_, P, OFF, CT, CH) }

  int w = blockIdx.x;
  int tile = 0, ct = 0;
  WINO_ITEM(w, tile, ct)
  unsigned doff[IT];
  WINO_DOFF(tile, doff)
  if (w < nwork) WINO_ISSUE(0, doff, ct, 0)
  int par = 0;
  bool first = true;

  // activation: none or ReLU (all the 3x3 s1 layers of the network), branch-free as max(v, lo)
  const float act_lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();
  const bool has_res = a.res != nullptr;
  const unsigned rowpitch = (unsigned)(a.Wo * Co) * 4u, colpitch = (unsigned)Co * 4u;

  for (; w < nwork; w += gridDim.x) {
    // next item (its first stage is prefetched during this item's last K step) and this item's outputs
    int tile_n = 0, ct_n = 0;
    WINO_ITEM(w + (int)gridDim.x, tile_n, ct_n)
    const bool more = (w + (int)gridDim.x) < nwork;
    unsigned doff_n[IT];
    WINO_DOFF(tile_n, doff_n)
    unsigned voff[4];
    {
      const int tb_ = tile / tiles_xy;
      const int r_ = tile - tb_ * tiles_xy;
      const int ty_ = r_ / a.tiles_x, tx_ = r_ - ty_ * a.tiles_x;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tb_ * TNB + (og[r] >> 16);
        const int oy = ty_ * TH + 2 * ((og[r] >> 8) & 255), ox = tx_ * TW + 2 * (og[r] & 255);
        voff[r] = (tile < ntile && n < a.N && oy < a.Ho && ox < a.Wo)
                      ? (unsigned)(((n * a.Ho + oy) * a.Wo + ox) * Co + ct * WN_CO + li) * 4u
                      : EGN_OOB;
      }
    }
    float sc[WN_NT], sh[WN_NT];
#pragma unroll
    for (int nt = 0; nt < WN_NT; ++nt) {
      sc[nt] = a.scale[ct * WN_CO + nt * 16 + li];
      sh[nt] = a.shift[ct * WN_CO + nt * 16 + li];
    }

    f32x4 acc[16][WN_NT];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
      for (int nt = 0; nt < WN_NT; ++nt) acc[f][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rv[4][2][2][WN_NT];  // residual of this lane's outputs: [tile r][a][b][nt]

    for (int c = 0; c < nchunk; ++c) {
      const bool last = c + 1 == nchunk;
      // stage `par` (halo chunk c + U slab) must have landed in every wave's share; the only younger
      // vector-memory operations are the 48 stores of the previous item's epilogue (they may stay in flight)
      asm volatile("" ::: "memory");
      if constexpr (!(ABL & 8)) {
        if (c == 0 && !first) __builtin_amdgcn_s_waitcnt(0xC070);  // vmcnt(48) expcnt(7) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0)
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
      first = false;
      // the next stage: (this item, chunk c+1) or (next item, chunk 0); its NPIECE DMA instructions are
      // issued one by one between the frequency groups below (a burst of 19 at the top cost 14 % of a step)
      const bool nx_issue = !(ABL & 1) && (!last || more);
      const int nx_ct = last ? ct_n : ct, nx_ch = last ? 0 : c + 1;
      unsigned nxo[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) nxo[it] = last ? doff_n[it] : doff[it];
#define WINO_NEXT(K)                                     \
  {                                                      \
    __builtin_amdgcn_sched_barrier(0x0106);              \
    if (nx_issue) WINO_PIECE(K, par ^ 1, nxo, nx_ct, nx_ch) \
    __builtin_amdgcn_sched_barrier(0x0106);              \
  }
      WINO_NEXT(0) WINO_NEXT(1) WINO_NEXT(2)
      static_assert(NPIECE <= 19, "3 DMA pieces at the top + one per frequency");
      if (last) {
        // the vmcnt(48) above counts on program order DMA -> residual loads -> stores: the loads go
        // after the LAST piece (end of this step); here only the offsets are prepared
      }

      // ---- input transform V = B^T d B of this lane's (tile, channel quad), row by row ----
      const float4* hb01 = sH + par * D::BUF + pb01;
      const float4* hb23 = sH + par * D::BUF + pb23;
      f32x4 d[4][4], t[4][4], V[16];
#define WINO_LOADROW(R)                                                                                   \
  _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                                      \
    if constexpr ((ABL & 2)) d[R][cc] = f32x4{(float)lane, (float)((R) + c), (float)cc, 1.f};              \
    else d[R][cc] = *reinterpret_cast<const f32x4*>(&((R) < 2 ? hb01 : hb23)[(R)*G::ROFF + cc]);          \
  }
#define WINO_VROW(I)                          \
  {                                           \
    V[(I)*4 + 0] = t[I][0] - t[I][2];         \
    V[(I)*4 + 1] = t[I][1] + t[I][2];         \
    V[(I)*4 + 2] = t[I][2] - t[I][1];         \
    V[(I)*4 + 3] = t[I][1] - t[I][3];         \
  }
      WINO_LOADROW(0) WINO_LOADROW(2)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) t[0][cc] = d[0][cc] - d[2][cc];
      WINO_VROW(0)
      WINO_LOADROW(1)
      WINO_LOADROW(3)

      // ---- 16 frequencies x 3 co sub-tiles x 4 k-steps ----
      const float4* ub = sU + par * WN_USLOTS + kq * WN_CO + li;
      f32x4 bf[2][WN_NT];
#pragma unroll
      for (int nt = 0; nt < WN_NT; ++nt) bf[0][nt] = *reinterpret_cast<const f32x4*>(&ub[nt * 16]);
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        if (f + 1 < 16) {
#pragma unroll
          for (int nt = 0; nt < WN_NT; ++nt)
            bf[(f + 1) & 1][nt] = *reinterpret_cast<const f32x4*>(&ub[(f + 1) * EGN_CKQ * WN_CO + nt * 16]);
        }
        // the next frequency row's V: in source order here so that its adds land between the MFMAs
        if (f == 0) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            t[1][cc] = d[1][cc] + d[2][cc];
            t[2][cc] = d[2][cc] - d[1][cc];
          }
          WINO_VROW(1)
        } else if (f == 4) {
          WINO_VROW(2)
        } else if (f == 8) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) t[3][cc] = d[1][cc] - d[3][cc];
          WINO_VROW(3)
        }
        // k-step outermost: consecutive MFMAs hit different accumulators (40-cycle dependent latency
        // vs 32-cycle issue -- one wave per SIMD has no partner to fill the bubble)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nt = 0; nt < WN_NT; ++nt)
            acc[f][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[f][s], bf[f & 1][nt][s], acc[f][nt], 0, 0, 0);
        if (3 + f < NPIECE) WINO_NEXT(3 + f)
      }
#undef WINO_NEXT
#undef WINO_LOADROW
#undef WINO_VROW
      if (last) {
#pragma unroll
        for (int it = 0; it < IT; ++it) doff[it] = doff_n[it];
        // pin the program order DMA -> residual loads (the asm has no memory clobber)
        asm volatile("" ::: "memory");
        // residual values (without a residual: OOB offsets, zeros, same instruction stream)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned ro = (has_res && !(ABL & 4)) ? voff[r] : EGN_OOB;
#pragma unroll
          for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
              for (int nt = 0; nt < WN_NT; ++nt)
                rv[r][pa][pb][nt] = wino_load4(rr, ro, pa * rowpitch + pb * colpitch + nt * 64u);
        }
      }
      par ^= 1;
    }

    // ---- output transform Y = A^T M A, epilogue, stores ----
#pragma unroll
    for (int nt = 0; nt < WN_NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float tt[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float m0 = acc[i * 4 + 0][nt][r], m1 = acc[i * 4 + 1][nt][r];
          const float m2 = acc[i * 4 + 2][nt][r], m3 = acc[i * 4 + 3][nt][r];
          tt[i][0] = m0 + m1 + m2;
          tt[i][1] = m1 - m2 - m3;
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const float y0 = tt[0][pb] + tt[1][pb] + tt[2][pb];
          const float y1 = tt[1][pb] - tt[2][pb] - tt[3][pb];
#pragma unroll
          for (int pa = 0; pa < 2; ++pa) {
            float v = (pa == 0 ? y0 : y1) * sc[nt] + sh[nt] + rv[r][pa][pb][nt];
            v = fmaxf(v, act_lo);
            wino_store4(ry, ((ABL & 4) && v != 12345.678f) ? EGN_OOB : voff[r], pa * rowpitch + pb * colpitch + nt * 64u, v);
          }
        }
      }
    tile = tile_n;
    ct = ct_n;
  }
#undef WINO_ITEM
#undef WINO_DOFF
#undef WINO_PIECE
#undef WINO_ISSUE
}

// ---------------------------------------------------------------------------------------------
// conv_wino8_kernel -- the same algorithm with EIGHT waves (two per SIMD).
//
// What the 4-wave kernel loses (ablations, profiles/r2_wino_ablation.txt): with one wave per SIMD every
// DMA issue (~19 per K step), every ds_read wait and the input transform stall the matrix pipe, because
// nothing else can issue MFMAs on that SIMD -- 55..65 us where the MFMA + B-read loop alone takes 40.
// 192 accumulator registers per wave forbid a second wave.  So the 16 frequencies of an m-tile are
// split between TWO waves on the same tile: wave (mt, fh) owns frequency rows {2fh, 2fh+1} -- 8
// frequencies, 96 accumulators, ~230 registers -> 2 waves per SIMD, and one wave's DMA / transform /
// waits hide under its partner's MFMAs.  Per K step a wave reads 3 patch rows (12 ds_read_b128),
// forms its two rows of V (8 of the 16 values), reads 24 B fragments, issues 96 MFMAs and 9-10 DMAs.
// The output transform needs all four frequency rows: Y0 = t0 + t1 + t2, Y1 = t1 - t2 - t3 with
// t_i = (row i of M) A.  Wave fh = 0 keeps t0 + t1 (its share of Y0) and SENDS t1 (Y1's); wave
// fh = 1 keeps -t2 - t3 and sends t2 -- 24 floats per lane through LDS, in the U stage buffer the
// last K step has just consumed (8 waves x 6 KB = exactly its 48 KB), between two barriers; then
// each wave finishes and stores the output rows a = fh of its tile.  Per item: nchunk + 2 barriers.
// (ABL bit 4, this kernel only: the input transform of chunk 0 is re-used by every K step -- an upper
//  bound on what hiding the step-start patch reads + transform would buy: 61 -> 57, 56 -> 51, 51 -> 46 us)
// NW = 8 waves on 64 tiles, or NW = 4 waves on 32 tiles (two 8 x 8 images): the 8 x 8 maps at batch 64
// are only 128 (tile, co-tile) items in the 64-tile form -- half the CUs idle -- and 256 in this one.
template <int TH, int TW, int TNB, int ABL = 0, int NW = 8, int NT = 3>
__global__ __launch_bounds__(64 * NW, 2) void conv_wino8_kernel(ConvArgs a) {
  constexpr int CO_T = 16 * NT;                             // output channels per block
  constexpr int USL = 16 * EGN_CKQ * CO_T;                  // float4 per (chunk, co-tile) slab of U
  using G = WinoGeom<TH, TW, TNB>;
  constexpr int NTH = 64 * NW;
  constexpr int MTILES = NW / 2;
  static_assert(TNB * (TH / 2) * (TW / 2) == 16 * MTILES, "one m-tile per wave pair");
  constexpr int SLOTS = EGN_CKQ * G::PLANE;                 // multiple of 64: whole waves
  constexpr int IT = (SLOTS + NTH - 1) / NTH;               // halo DMA instructions (the last one partial)
  constexpr int BUF = SLOTS;
  constexpr int UIT = USL / NTH;                      // 6 (12 with 4 waves)
  constexpr int NPIECE = IT + UIT;                          // 10 (16)
  constexpr int TOPP = 2;                                   // DMA pieces issued at the top of a K step ...
  constexpr int PERF = (NPIECE - TOPP + 7) / 8;             // ... and after each of the 8 frequencies
  static_assert(SLOTS % 64 == 0 && USL % NTH == 0, "whole-wave DMA pieces");
  static_assert(NW * 2 * NT * 64 <= USL, "the partial exchange fits in one U stage buffer");
  extern __shared__ float4 smem[];
  float4* sU = smem;                  // [2][USL]
  float4* sH = smem + 2 * USL;  // [2][BUF]
  double* sS = reinterpret_cast<double*>(smem + 2 * USL + 2 * BUF);  // [NW][2][CO_T] BatchNorm partial sums
  // ABL & 32 (tools/wino_clk.py only): s_memtime stamps of every wave's lane 0 -- kept in LDS, copied to the
  // buffer passed as `res` at the end of the kernel (no residual then), so the vmcnt bookkeeping is untouched
  unsigned long long* sT = reinterpret_cast<unsigned long long*>(sS + NW * 2 * CO_T);   // [NW][W8_NTK]
  constexpr int W8_NTK = 48;
  int ntk = 0;
#define W8_CLK()                                                                     \
  {                                                                                  \
    if constexpr ((ABL & 32) != 0) {                                                 \
      if (lane == 0 && ntk < W8_NTK) sT[wave * W8_NTK + ntk] = __builtin_readcyclecounter(); \
      ++ntk;                                                                         \
    }                                                                                \
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave % MTILES;   // m-tile (16 Winograd tiles)
  const int fh = wave / MTILES;   // frequency rows {2fh, 2fh+1}
  const int li = lane & 15;
  const int kq = lane >> 4;

  const int C = a.Cin, Co = a.Cout;
  const int nct = Co / CO_T;
  const int nchunk = a.nchunk;

  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const unsigned long long uaddr = reinterpret_cast<unsigned long long>(a.w);
  const u32x4 rxv = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu,
                     (unsigned)((size_t)a.N * a.H * a.W * C * 4), 0x00020000u};
  const u32x4 ruv = {(unsigned)uaddr, (unsigned)(uaddr >> 32) & 0xffffu,
                     (unsigned)((size_t)nct * nchunk * USL * 16), 0x00020000u};
  const unsigned out_bytes = (unsigned)((size_t)a.N * a.Ho * a.Wo * Co * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, out_bytes, 0x00020000);

  int hmeta[IT];
  unsigned hq[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * NTH + tid;
    const int q = e / G::PLANE;
    hmeta[it] = (e < SLOTS && q < EGN_CKQ) ? G::decode(e - q * G::PLANE) : -1;
    hq[it] = (unsigned)q * 16u;
  }
  // patch rows fh, fh+1, fh+2 of this lane's tile: float4 index of column 0 in a halo buffer
  int prow[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int r = fh + k;
    prow[k] = kq * G::PLANE + G::patch_base(mt, li, r >> 1) + r * G::ROFF;
  }
  int og[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) og[r] = G::out_tile(mt, 4 * kq + r);

  const int tiles_xy = a.tiles_x * a.tiles_y;
  const int ntile = tiles_xy * ((a.N + TNB - 1) / TNB);
  const int nwork = ((ntile + 7) >> 3) * nct * 8;

#define W8_ITEM(Wi, TILE_, CT_)              \
  {                                          \
    const int x_ = (Wi)&7, q_ = (Wi) >> 3;   \
    TILE_ = (q_ / nct) * 8 + x_;             \
    CT_ = q_ - (q_ / nct) * nct;             \
  }
#define W8_DOFF(TILE_, OUT)                                                                         \
  {                                                                                                 \
    const int tb_ = (TILE_) / tiles_xy;                                                             \
    const int r_ = (TILE_)-tb_ * tiles_xy;                                                          \
    const int ty_ = r_ / a.tiles_x, tx_ = r_ - ty_ * a.tiles_x;                                     \
    const int n0_ = tb_ * TNB, iy0_ = ty_ * TH - 1, ix0_ = tx_ * TW - 1;                            \
    _Pragma("unroll") for (int it = 0; it < IT; ++it) {                                             \
      const int m_ = hmeta[it];                                                                     \
      const int n_ = n0_ + (m_ >> 16), iy_ = iy0_ + ((m_ >> 8) & 255), ix_ = ix0_ + (m_ & 255);     \
      const bool in_ = m_ >= 0 && (TILE_) < ntile && n_ < a.N && iy_ >= 0 && iy_ < a.H && ix_ >= 0 && ix_ < a.W; \
      OUT[it] = in_ ? (unsigned)(((n_ * a.H + iy_) * a.W + ix_) * C) * 4u + hq[it] : EGN_OOB;       \
    }                                                                                               \
  }
// DMA instruction K (0 .. NPIECE-1) of a K step; a wave beyond the end of the halo image skips its
// share of the last halo piece (wave-uniform)
#define W8_PIECE(K, P, OFF, CT, CH)                                                                  \
  {                                                                                                  \
    if ((K) < IT) {                                                                                  \
      if ((K)*NTH + wave * 64 < SLOTS)                                                               \
        wino_dma16(rxv, wino_lds_addr(sH + (P)*BUF + wave * 64) + (K)*NTH * 16, OFF[(K) < IT ? (K) : 0], \
                   (unsigned)(CH)*64u);                                                              \
    } else {                                                                                         \
      wino_dma16(ruv, wino_lds_addr(sU + (P)*USL + wave * 64) + ((K)-IT) * NTH * 16, (unsigned)tid * 16u, \
                 (unsigned)(((CT)*nchunk + (CH)) * USL) * 16u + ((K)-IT) * NTH * 16);          \
    }                                                                                                \
  }

  int w = blockIdx.x;
  int tile = 0, ct = 0;
  W8_ITEM(w, tile, ct)
  const int ct_block = ct;            // constant over the block's items (the launcher sizes the grid for it)
  if (a.stats != nullptr) {           // own-wave rows only: no barrier needed before the first use
    for (int e = lane; e < 2 * CO_T; e += 64) sS[wave * 2 * CO_T + e] = 0.0;
  }
  unsigned doff[IT];
  W8_DOFF(tile, doff)
  if (w < nwork) {
#pragma unroll
    for (int k_ = 0; k_ < NPIECE; ++k_) W8_PIECE(k_, 0, doff, ct, 0)
  }
  int par = 0;
  bool first = true;

  const float act_lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();
  const bool has_res = (ABL & 32) ? false : a.res != nullptr;
  const unsigned rowpitch = (unsigned)(a.Wo * Co) * 4u, colpitch = (unsigned)Co * 4u;
  W8_CLK()   // 0: kernel start (prologue DMA issued)

  for (; w < nwork; w += gridDim.x) {
    int tile_n = 0, ct_n = 0;
    W8_ITEM(w + (int)gridDim.x, tile_n, ct_n)
    const bool more = (w + (int)gridDim.x) < nwork;
    unsigned doff_n[IT];
    W8_DOFF(tile_n, doff_n)
    unsigned voff[4];   // this wave's output rows: a = fh of each of its 4 tiles
    {
      const int tb_ = tile / tiles_xy;
      const int r_ = tile - tb_ * tiles_xy;
      const int ty_ = r_ / a.tiles_x, tx_ = r_ - ty_ * a.tiles_x;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = tb_ * TNB + (og[r] >> 16);
        const int oy = ty_ * TH + 2 * ((og[r] >> 8) & 255) + fh, ox = tx_ * TW + 2 * (og[r] & 255);
        voff[r] = (tile < ntile && n < a.N && oy < a.Ho && ox < a.Wo)
                      ? (unsigned)(((n * a.Ho + oy) * a.Wo + ox) * Co + ct * CO_T + li) * 4u
                      : EGN_OOB;
      }
    }
    float sc[NT], sh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      sc[nt] = a.scale[ct * CO_T + nt * 16 + li];
      sh[nt] = a.shift[ct * CO_T + nt * 16 + li];
    }

    f32x4 acc[8][NT];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[f][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rv[4][2][NT];  // residual of this lane's outputs: [tile r][b][nt]
    f32x4 Vkeep[8];         // (ABL & 16 only: the transform of chunk 0 re-used by every K step)

    for (int c = 0; c < nchunk; ++c) {
      const bool last = c + 1 == nchunk;
      asm volatile("" ::: "memory");
      W8_CLK()   // step top
      if constexpr (!(ABL & 8)) {
        if (c == 0 && !first) __builtin_amdgcn_s_waitcnt(0x4078);  // vmcnt(24): all but the last item's stores
        else __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0)
        if constexpr ((ABL & 32) != 0) { asm volatile("" ::: "memory"); W8_CLK() asm volatile("" ::: "memory"); }  // own DMA landed
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
      W8_CLK()   // past the barrier
      first = false;
      const bool nx_issue = !(ABL & 1) && (!last || more);
      const int nx_ct = last ? ct_n : ct, nx_ch = last ? 0 : c + 1;
      unsigned nxo[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) nxo[it] = last ? doff_n[it] : doff[it];
#define W8_NEXT(K)                                        \
  {                                                       \
    __builtin_amdgcn_sched_barrier(0x0106);               \
    if (nx_issue) W8_PIECE(K, par ^ 1, nxo, nx_ct, nx_ch) \
    __builtin_amdgcn_sched_barrier(0x0106);               \
  }
      // ---- this wave's two rows of V = B^T d B ----
      const float4* hb = sH + par * BUF;
      f32x4 d[3][4], V[8];
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          if constexpr ((ABL & 2)) d[k][cc] = f32x4{(float)lane, (float)(k + c), (float)cc, 1.f};
          else if ((ABL & 16) && c != 0) d[k][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
          else d[k][cc] = *reinterpret_cast<const f32x4*>(&hb[prow[k] + cc]);
        }
      W8_NEXT(0) W8_NEXT(1)
      static_assert(TOPP == 2 && NPIECE <= TOPP + 8 * PERF, "every DMA piece has a slot");
      {
        f32x4 ta[4], tb[4];
        // fh = 0: rows (d0, d1, d2): T0 = d0 - d2, T1 = d1 + d2;  fh = 1: rows (d1, d2, d3): T2 = d2 - d1, T3 = d1 - d3
        if (fh == 0) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            ta[cc] = d[0][cc] - d[2][cc];
            tb[cc] = d[1][cc] + d[2][cc];
          }
        } else {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            ta[cc] = d[1][cc] - d[0][cc];
            tb[cc] = d[0][cc] - d[2][cc];
          }
        }
        V[0] = ta[0] - ta[2]; V[1] = ta[1] + ta[2]; V[2] = ta[2] - ta[1]; V[3] = ta[1] - ta[3];
        V[4] = tb[0] - tb[2]; V[5] = tb[1] + tb[2]; V[6] = tb[2] - tb[1]; V[7] = tb[1] - tb[3];
      }
      if constexpr ((ABL & 16) != 0) {  // timing experiment: what would hiding the step-start transform buy
        if (c == 0) {
#pragma unroll
          for (int f = 0; f < 8; ++f) Vkeep[f] = V[f];
        } else {
#pragma unroll
          for (int f = 0; f < 8; ++f) V[f] = Vkeep[f];
        }
      }

      // ---- 8 frequencies x 3 co sub-tiles x 4 k-steps ----
      const float4* ub = sU + par * USL + (fh * 8 * EGN_CKQ + kq) * CO_T + li;
      f32x4 bf[2][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *reinterpret_cast<const f32x4*>(&ub[nt * 16]);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        if (f + 1 < 8) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            bf[(f + 1) & 1][nt] = *reinterpret_cast<const f32x4*>(&ub[(f + 1) * EGN_CKQ * CO_T + nt * 16]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[f][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[f][s], bf[f & 1][nt][s], acc[f][nt], 0, 0, 0);
#pragma unroll
        for (int k_ = 0; k_ < PERF; ++k_)
          if (TOPP + f * PERF + k_ < NPIECE) W8_NEXT(TOPP + f * PERF + k_)
      }
#undef W8_NEXT
      if (last) {
#pragma unroll
        for (int it = 0; it < IT; ++it) doff[it] = doff_n[it];
        asm volatile("" ::: "memory");  // program order DMA -> residual loads (the vmcnt(24) above counts on it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned ro = (has_res && !(ABL & 4)) ? voff[r] : EGN_OOB;
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) rv[r][pb][nt] = wino_load4(rr, ro, pb * colpitch + nt * 64u);
        }
      }
      par ^= 1;
    }

    W8_CLK()   // K loop done
    // ---- output transform: this wave's frequency rows, then the exchange with the partner wave ----
    float keep[NT][4][2], send[NT][4][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          float t0, t1;  // t of the wave's first / second frequency row, column pb of A
          if (pb == 0) {
            t0 = acc[0][nt][r] + acc[1][nt][r] + acc[2][nt][r];
            t1 = acc[4][nt][r] + acc[5][nt][r] + acc[6][nt][r];
          } else {
            t0 = acc[1][nt][r] - acc[2][nt][r] - acc[3][nt][r];
            t1 = acc[5][nt][r] - acc[6][nt][r] - acc[7][nt][r];
          }
          // fh = 0: rows 0, 1: keep t0 + t1 (Y0), send t1 (Y1);  fh = 1: rows 2, 3: keep -t2 - t3 (Y1), send t2 (Y0)
          keep[nt][r][pb] = fh == 0 ? t0 + t1 : -t0 - t1;
          send[nt][r][pb] = fh == 0 ? t1 : t0;
        }
    {
      // the U buffer of the last K step (par ^ 1 after the toggle) is free once every wave is past its
      // MFMAs: barrier, write, barrier, read the partner's
      float4* xch = sU + (par ^ 1) * USL;
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k4 = 0; k4 < 2 * NT; ++k4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = k4 * 4 + e;  // (nt*4 + r)*2 + pb
          v[e] = send[idx >> 3][(idx >> 1) & 3][idx & 1];
        }
        *reinterpret_cast<f32x4*>(&xch[(wave * 2 * NT + k4) * 64 + lane]) = v;
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k4 = 0; k4 < 2 * NT; ++k4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&xch[((wave ^ MTILES) * 2 * NT + k4) * 64 + lane]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = k4 * 4 + e;
          keep[idx >> 3][(idx >> 1) & 3][idx & 1] += v[e];
        }
      }
    }
    W8_CLK()   // exchange done
    float st1[NT], st2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) st1[nt] = st2[nt] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          float v = keep[nt][r][pb] * sc[nt] + sh[nt] + rv[r][pb][nt];
          v = fmaxf(v, act_lo);
          wino_store4(ry, ((ABL & 4) && v != 12345.678f) ? EGN_OOB : voff[r], pb * colpitch + nt * 64u, v);
          if (voff[r] != EGN_OOB) { st1[nt] += v; st2[nt] += v * v; }
        }
    if (a.stats != nullptr) {
      // BatchNorm batch statistics ride in the epilogue (training: raw conv output, scale 1 / shift 0):
      // this wave's column sums over its 16 tiles x 2 pixels -- lanes li + 16*kq hold the same channel --
      // are added to the wave's own row of a small LDS table (doubles); the block writes ONE partial
      // row at the end of the kernel (its co-tile is the same for all its items, see wino8_grid)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        st1[nt] += __shfl_xor(st1[nt], 16);
        st1[nt] += __shfl_xor(st1[nt], 32);
        st2[nt] += __shfl_xor(st2[nt], 16);
        st2[nt] += __shfl_xor(st2[nt], 32);
      }
      if (kq == 0) {
        double* srow = sS + wave * 2 * CO_T + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          srow[nt * 16] += (double)st1[nt];
          srow[CO_T + nt * 16] += (double)st2[nt];
        }
      }
    }
    W8_CLK()   // stores issued
    tile = tile_n;
    ct = ct_n;
  }
  if constexpr ((ABL & 32) != 0) {
    __syncthreads();
    unsigned long long* out = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res)) +
                              (size_t)blockIdx.x * (NW * W8_NTK + 1);
    for (int e = tid; e < NW * W8_NTK; e += NTH) out[1 + e] = sT[e];
    if (tid == 0) out[0] = (unsigned long long)ntk;
  }
  if (a.stats != nullptr) {
    // one partial row per block: [2][Cout] doubles, the block's 48 columns = its waves' sums in wave
    // order, zeros elsewhere (egn_bn_stats_finalize_f32 adds the rows of all blocks in a fixed order)
    __syncthreads();
    double* row = a.stats + (size_t)blockIdx.x * 2 * Co;
    for (int e = tid; e < 2 * Co; e += NTH) {
      const int which = e / Co, c = e - which * Co;
      const int cl = c - ct_block * CO_T;
      double v = 0.0;
      if (cl >= 0 && cl < CO_T) {
        for (int k = 0; k < NW; ++k) v += sS[(k * 2 + which) * CO_T + cl];
      }
      row[e] = v;
    }
  }
#undef W8_ITEM
#undef W8_DOFF
#undef W8_PIECE
#undef W8_CLK
}

// ---------------------------------------------------------------------------------------------
// conv_wino9_kernel -- conv_wino8_kernel with the VALU instruction count cut in half [round 3].
//
// What round 3 measured (profiles/r3_mfma_tax.txt, a hand-written v_mfma_f32_16x16x4_f32 stream with
// companions): beside fp32 MFMAs a ds_read_b128, an LDS-DMA piece, a global load and an s_barrier cost
// ~0-3 cycles of matrix-pipe time each -- but every plain VALU instruction costs 5-10 (one wave per SIMD) /
// 3-6 (two) cycles: the fp32 MFMA executes on the SIMD's fp32 FMA lanes, a VALU instruction of either wave
// takes issue slots away from it.  conv_wino8_kernel issues 3.2 VALU instructions per MFMA on the 48-channel
// layers (928 per item, 288 MFMAs; ISA count) of which 1.4 are arithmetic the algorithm needs; its timeline
// (profiles/r3_wino8_timeline_before.txt) shows a K step at 7 300-7 500 cycles for 6 144 of MFMA issue and
// 20 % of an item in the head / exchange / epilogue phases.  This kernel keeps the data flow and removes
// instructions:
//   * item index arithmetic on the SCALAR unit: divisions by multiply-high with launcher-provided
//     multipliers on readfirstlane'd values; per-lane halo / output offsets = uniform base + a per-lane
//     relative offset computed once per kernel, validity by unsigned compares (6 VALU per offset);
//   * ONE instruction stream for both frequency halves: wave fh = 1 owns its two frequency rows in swapped
//     order (row 3, row 2), the patch rows each role reads are per-wave LDS offsets, the one sign that
//     differs is a wave-uniform multiplier in a v_fma -- no branches, no register shuffles (the branchy
//     form cost 29 v_mov per K step), `keep = t0 + t1`, `send = t1` for both;
//   * transforms as single v_add / v_sub / v_fma through asm wrappers: the compiler's SLP pass packed them
//     into v_pk_add_f32, which costs more than two plain adds beside MFMAs (MI355X_MICROARCH.md);
//   * the next stage's DMA pieces all go out in the first quarter of a K step (they cost nothing to issue;
//     spread over the step the last ones had no time to land before the barrier).
// Same LDS layout, same filter packing (the U rows of wave fh = 1 are read in swapped order), same
// exchange, same BatchNorm statistics, same results bit for bit as conv_wino8_kernel.
__device__ __forceinline__ float wn_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float wn_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float wn_fma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float wn_fma_s(float s, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(s), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float wn_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x4 wn_sub4(f32x4 a, f32x4 b) { return f32x4{wn_sub(a[0], b[0]), wn_sub(a[1], b[1]), wn_sub(a[2], b[2]), wn_sub(a[3], b[3])}; }
__device__ __forceinline__ f32x4 wn_add4(f32x4 a, f32x4 b) { return f32x4{wn_add(a[0], b[0]), wn_add(a[1], b[1]), wn_add(a[2], b[2]), wn_add(a[3], b[3])}; }
__device__ __forceinline__ f32x4 wn_fma4_s(float s, f32x4 b, f32x4 c) { return f32x4{wn_fma_s(s, b[0], c[0]), wn_fma_s(s, b[1], c[1]), wn_fma_s(s, b[2], c[2]), wn_fma_s(s, b[3], c[3])}; }
__device__ __forceinline__ unsigned wn_udiv(unsigned x, unsigned magic) { return magic ? __umulhi(x, magic) : x; }

// KQ = channel quads per K stage: 4 (a whole 16-channel chunk of the packed filter, one 140 KB block per CU) or
// 2 (half a chunk: 8-channel stages, 68 KB of LDS -- TWO 4-wave blocks per CU, one wave of each on every SIMD.
// The blocks run out of phase: while one sits in its barrier, its exchange or its epilogue the other one's
// MFMAs keep the matrix pipe busy -- profiles/r3_wino8_vs_wino9_timeline.txt has the lock-stepped 8-wave block
// at 8 400 cycles per 16 channels against 6 144 of MFMA issue, plus 4 400 cycles of MFMA-free phases per item).
// With KQ = 2 lane group kq reads quad kq & 1 and the float pair kq >> 1 of it: the MFMA's four k lanes cover
// channels {0, 1} of both quads, then {2, 3}: KQ MFMAs per frequency and co sub-tile, the same VALU work per MFMA.
template <int KQ> struct WinoVec;
template <> struct WinoVec<4> { typedef f32x4 type; };
template <> struct WinoVec<2> { typedef float __attribute__((ext_vector_type(2))) type; };

template <int TH, int TW, int TNB, int NW = 8, int NT = 3, int CLK = 0, int KQ = EGN_CKQ>
__global__ __launch_bounds__(64 * NW, 2) void conv_wino9_kernel(ConvArgs a) {
  constexpr int CO_T = 16 * NT;
  constexpr int USL = 16 * KQ * CO_T;           // float4 slots of one filter stage
  constexpr int HS = EGN_CKQ / KQ;              // stages per 16-channel chunk
  constexpr int UCH = 16 * EGN_CKQ * CO_T;      // float4 slots of one packed chunk
  typedef typename WinoVec<KQ>::type vk_t;
  using G = WinoGeom<TH, TW, TNB>;
  constexpr int NTH = 64 * NW;
  constexpr int MTILES = NW / 2;
  static_assert(TNB * (TH / 2) * (TW / 2) == 16 * MTILES, "one m-tile per wave pair");
  constexpr int SLOTS = KQ * G::PLANE;
  constexpr int IT = (SLOTS + NTH - 1) / NTH;
  constexpr int BUF = SLOTS;
  constexpr int UIT = USL / NTH;
  constexpr int NPIECE = IT + UIT;
  static_assert(SLOTS % 64 == 0 && USL % NTH == 0, "whole-wave DMA pieces");
  static_assert(NW * 2 * NT * 64 <= USL, "the partial exchange fits in one U stage buffer");
  extern __shared__ float4 smem[];
  float4* sU = smem;
  float4* sH = smem + 2 * USL;
  double* sS = reinterpret_cast<double*>(smem + 2 * USL + 2 * BUF);
  // CLK (tools/wino_clk.py only): s_memtime stamps like conv_wino8_kernel's ABL & 32 build, `res` = the stamp buffer
  unsigned long long* sT = reinterpret_cast<unsigned long long*>(sS + NW * 2 * CO_T);
  constexpr int W9_NTK = 48;
  int ntk = 0;
#define W9_CLK()                                                                     \
  {                                                                                  \
    if constexpr (CLK != 0) {                                                        \
      if (lane == 0 && ntk < W9_NTK) sT[wave * W9_NTK + ntk] = __builtin_readcyclecounter(); \
      ++ntk;                                                                         \
    }                                                                                \
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave % MTILES;
  const int fh = wave / MTILES;
  const int li = lane & 15;
  const int kq = lane >> 4;
  const int kqq = kq % KQ, kqc = (kq / KQ) * KQ;   // this lane's quad of the stage, first float of its share
  const float sigma = fh == 0 ? 1.f : -1.f;     // wave-uniform sign of the one term that differs between the halves

  const int C = a.Cin, Co = a.Cout;
  const int nct = Co / CO_T;
  const int nchunk = a.nchunk;

  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const unsigned long long uaddr = reinterpret_cast<unsigned long long>(a.w);
  const u32x4 rxv = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu,
                     (unsigned)((size_t)a.N * a.H * a.W * C * 4), 0x00020000u};
  const u32x4 ruv = {(unsigned)uaddr, (unsigned)(uaddr >> 32) & 0xffffu,
                     (unsigned)((size_t)nct * nchunk * UCH * 16), 0x00020000u};
  const unsigned out_bytes = (unsigned)((size_t)a.N * a.Ho * a.Wo * Co * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, out_bytes, 0x00020000);

  // ---- per-lane, item-independent: halo slot -> (image, hy, hx) and its byte offset relative to the tile's
  // origin pixel (n0, iy0, ix0); the tile origin is a wave-uniform base added per item
  int hyx[IT];        // hy << 16 | hx, -1 = pad slot
  int hb[IT];         // image within the tile batch (TNB > 1)
  int hrel[IT];       // ((hb * H + hy) * W + hx) * C * 4 + quad * 16
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = it * NTH + tid;
    const int q = e / G::PLANE;
    const int m = (e < SLOTS && q < KQ) ? G::decode(e - q * G::PLANE) : -1;
    const int b_ = m >> 16, y_ = (m >> 8) & 255, x_ = m & 255;
    hyx[it] = m < 0 ? -1 : ((y_ << 16) | x_);
    hb[it] = m < 0 ? 0 : b_;
    hrel[it] = m < 0 ? 0 : ((b_ * a.H + y_) * a.W + x_) * C * 4 + q * 16;
  }
  // patch rows of this lane's tile; roles (X, Y, Z, W): ta = X - Y, tb = Z + sigma * W
  //   fh = 0 (rows 0, 1):  T0 = d0 - d2,  T1 = d1 + d2      X = 0, Y = 2, Z = 1, W = 2
  //   fh = 1 (rows 3, 2):  T3 = d1 - d3,  T2 = d2 - d1      first row = T3: X = 1, Y = 3;  second = T2 = d2 + (-1) d1: Z = 2, W = 1
  int prow[4];
  {
    int pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pr[r] = kqq * G::PLANE + G::patch_base(mt, li, r >> 1) + r * G::ROFF;
    prow[0] = fh == 0 ? pr[0] : pr[1];
    prow[1] = fh == 0 ? pr[2] : pr[3];
    prow[2] = fh == 0 ? pr[1] : pr[2];
    prow[3] = fh == 0 ? pr[2] : pr[1];
  }
  // this wave's two frequency rows in U: first = row (fh ? 3 : 0), second = row (fh ? 2 : 1)
  const int urow0 = (fh == 0 ? 0 : 3) * 4, urow1 = (fh == 0 ? 1 : 2) * 4;
  // output pixels of this lane: 4 tiles (MFMA result rows 4kq + r), output row a = fh of each
  int oyx[4], ob[4], orel[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int og = G::out_tile(mt, 4 * kq + r);
    const int b_ = og >> 16, y_ = 2 * ((og >> 8) & 255) + fh, x_ = 2 * (og & 255);
    oyx[r] = (y_ << 16) | x_;
    ob[r] = b_;
    orel[r] = (((b_ * a.Ho + y_) * a.Wo + x_) * Co + li) * 4;
  }

  const int tiles_xy = a.tiles_x * a.tiles_y;
  const int ntile = tiles_xy * ((a.N + TNB - 1) / TNB);
  const int nwork = ((ntile + 7) >> 3) * nct * 8;
  const int gsz = __builtin_amdgcn_readfirstlane((int)gridDim.x);

  // wave-uniform item decode (scalar unit): w -> (tile, ct) -> (tb, ty, tx)
#define W9_ITEM(Wi, TILE_, CT_, TB_, TY_, TX_)                                   \
  {                                                                              \
    const unsigned wi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(Wi));    \
    const unsigned x_ = wi_ & 7u, q_ = wi_ >> 3;                                 \
    const unsigned qq_ = wn_udiv(q_, a.mg_nct);                                  \
    TILE_ = (int)(qq_ * 8u + x_);                                                \
    CT_ = (int)(q_ - qq_ * (unsigned)nct);                                       \
    const unsigned tb_ = wn_udiv((unsigned)TILE_, a.mg_txy);                     \
    const unsigned r_ = (unsigned)TILE_ - tb_ * (unsigned)tiles_xy;              \
    const unsigned ty_ = wn_udiv(r_, a.mg_tx);                                   \
    TB_ = (int)tb_; TY_ = (int)ty_; TX_ = (int)(r_ - ty_ * (unsigned)a.tiles_x); \
  }
  // halo DMA byte offsets (chunk 0) of a tile: uniform base + per-lane relative offset; pad slots and
  // out-of-image pixels get the OOB offset (the DMA writes zeros there = the convolution's zero padding)
#define W9_DOFF(TILE_, TB_, TY_, TX_, OUT)                                                                   \
  {                                                                                                          \
    const int n0_ = (TB_)*TNB, iy0_ = (TY_)*TH - 1, ix0_ = (TX_)*TW - 1;                                     \
    const int base_ = ((n0_ * a.H + iy0_) * a.W + ix0_) * C * 4;                                             \
    const bool tok_ = (TILE_) < ntile;                                                                       \
    _Pragma("unroll") for (int it = 0; it < IT; ++it) {                                                      \
      const unsigned iy_ = (unsigned)(iy0_ + (hyx[it] >> 16)), ix_ = (unsigned)(ix0_ + (hyx[it] & 0xffff));  \
      bool in_ = tok_ && hyx[it] >= 0 && iy_ < (unsigned)a.H && ix_ < (unsigned)a.W;                         \
      if (TNB > 1) in_ = in_ && (n0_ + hb[it]) < a.N;                                                        \
      OUT[it] = in_ ? (unsigned)(base_ + hrel[it]) : EGN_OOB;                                                \
    }                                                                                                        \
  }
  // CH = stage index: chunk CH / HS, quads (CH % HS) * KQ ... of it
#define W9_PIECE(K, P, OFF, CT, CH)                                                                  \
  {                                                                                                  \
    if ((K) < IT) {                                                                                  \
      if ((K)*NTH + wave * 64 < SLOTS)                                                               \
        wino_dma16(rxv, wino_lds_addr(sH + (P)*BUF + wave * 64) + (K)*NTH * 16, OFF[(K) < IT ? (K) : 0], \
                   (unsigned)(CH) * (KQ * 16u));                                                     \
    } else if constexpr (KQ == EGN_CKQ) {                                                            \
      wino_dma16(ruv, wino_lds_addr(sU + (P)*USL + wave * 64) + ((K)-IT) * NTH * 16, (unsigned)tid * 16u, \
                 (unsigned)(((CT)*nchunk + (CH)) * USL) * 16u + ((K)-IT) * NTH * 16);                \
    } else {                                                                                         \
      const unsigned ch_ = (unsigned)(CH) / HS, hs_ = (unsigned)(CH) - ch_ * HS;                     \
      wino_dma16(ruv, wino_lds_addr(sU + (P)*USL + wave * 64) + ((K)-IT) * NTH * 16, urel[(K) >= IT ? (K)-IT : 0], \
                 (((unsigned)(CT)*nchunk + ch_) * UCH + hs_ * (KQ * CO_T)) * 16u);                   \
    }                                                                                                \
  }
  // filter stage slot e = piece * NTH + tid -> frequency e / (KQ * CO_T): its KQ * CO_T slots sit EGN_CKQ * CO_T apart
  // in the packed chunk
  unsigned urel[KQ == EGN_CKQ ? 1 : UIT];
  if constexpr (KQ != EGN_CKQ) {
#pragma unroll
    for (int k_ = 0; k_ < UIT; ++k_) {
      const int e = k_ * NTH + tid, f_ = e / (KQ * CO_T);
      urel[k_] = (unsigned)(f_ * (EGN_CKQ * CO_T) + (e - f_ * (KQ * CO_T))) * 16u;
    }
  }

  int w = blockIdx.x;
  int tile = 0, ct = 0, tb = 0, ty = 0, tx = 0;
  W9_ITEM(w, tile, ct, tb, ty, tx)
  const int ct_block = ct;
  if (a.stats != nullptr) {
    for (int e = lane; e < 2 * CO_T; e += 64) sS[wave * 2 * CO_T + e] = 0.0;
  }
  unsigned doff[IT];
  W9_DOFF(tile, tb, ty, tx, doff)
  if (w < nwork) {
#pragma unroll
    for (int k_ = 0; k_ < NPIECE; ++k_) W9_PIECE(k_, 0, doff, ct, 0)
  }
  int par = 0;
  bool first = true;

  const float act_lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();
  const bool has_res = CLK ? false : a.res != nullptr;
  const unsigned rowpitch = (unsigned)(a.Wo * Co) * 4u, colpitch = (unsigned)Co * 4u;
  (void)rowpitch;
  W9_CLK()

  for (; w < nwork; w += gsz) {
    int tile_n = 0, ct_n = 0, tb_n = 0, ty_n = 0, tx_n = 0;
    W9_ITEM(w + gsz, tile_n, ct_n, tb_n, ty_n, tx_n)
    const bool more = (w + gsz) < nwork;
    unsigned doff_n[IT];
    W9_DOFF(tile_n, tb_n, ty_n, tx_n, doff_n)
    unsigned voff[4];
    {
      const int n0_ = tb * TNB, oy0_ = ty * TH, ox0_ = tx * TW;
      const int base_ = (((n0_ * a.Ho + oy0_) * a.Wo + ox0_) * Co + ct * CO_T) * 4;
      const bool tok_ = tile < ntile;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned oy_ = (unsigned)(oy0_ + (oyx[r] >> 16)), ox_ = (unsigned)(ox0_ + (oyx[r] & 0xffff));
        bool in_ = tok_ && oy_ < (unsigned)a.Ho && ox_ < (unsigned)a.Wo;
        if (TNB > 1) in_ = in_ && (n0_ + ob[r]) < a.N;
        voff[r] = in_ ? (unsigned)(base_ + orel[r]) : EGN_OOB;
      }
    }
    float sc[NT], sh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      sc[nt] = a.scale[ct * CO_T + nt * 16 + li];
      sh[nt] = a.shift[ct * CO_T + nt * 16 + li];
    }

    f32x4 acc[8][NT];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[f][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rv[4][2][NT];

    const int nstage = nchunk * HS;
    for (int c = 0; c < nstage; ++c) {
      const bool last = c + 1 == nstage;
      asm volatile("" ::: "memory");
      W9_CLK()
      if (c == 0 && !first) __builtin_amdgcn_s_waitcnt(0x4078);  // vmcnt(24): all but the last item's stores
      else __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0)
      if constexpr (CLK != 0) { asm volatile("" ::: "memory"); W9_CLK() asm volatile("" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      W9_CLK()
      first = false;
      const bool nx_issue = !last || more;
      const int nx_ct = last ? ct_n : ct, nx_ch = last ? 0 : c + 1;
      if (last) {        // from here on the pieces belong to the next item's first stage
#pragma unroll
        for (int it = 0; it < IT; ++it) doff[it] = doff_n[it];
      }
#define W9_NEXT(K)                                         \
  {                                                        \
    __builtin_amdgcn_sched_barrier(0x0106);                \
    if (nx_issue) W9_PIECE(K, par ^ 1, doff, nx_ct, nx_ch) \
    __builtin_amdgcn_sched_barrier(0x0106);                \
  }
      // ---- this wave's two rows of V = B^T d B: ta = X - Y, tb = Z + sigma W, then the column transform.
      // Plain arithmetic: the scheduler sinks each V row to its first use, so the MFMAs of row 0 start while row
      // 1 is still being formed -- asm wrappers pinned all 64 instructions in front of the first MFMA.  (The SLP
      // pass packs 6-8 of the 80 operations of a K step into v_pk_*_f32; forcing scalar instructions measured
      // no difference here or in conv_wino4.hip.)
      const float4* hb_ = sH + par * BUF;
      vk_t V[8];
      constexpr int TOPP = 2;                                   // DMA pieces issued at the top of a K step ...
      constexpr int PERF = (NPIECE - TOPP + 7) / 8;             // ... and after each of the 8 frequencies
      static_assert(NPIECE <= TOPP + 8 * PERF, "every DMA piece has a slot");
      {
        vk_t dx[4], dy[4], ta[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          dx[cc] = *reinterpret_cast<const vk_t*>(reinterpret_cast<const float*>(&hb_[prow[0] + cc]) + kqc);
          dy[cc] = *reinterpret_cast<const vk_t*>(reinterpret_cast<const float*>(&hb_[prow[1] + cc]) + kqc);
        }
        W9_NEXT(0) W9_NEXT(1)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) ta[cc] = dx[cc] - dy[cc];
        V[0] = ta[0] - ta[2]; V[1] = ta[1] + ta[2]; V[2] = ta[2] - ta[1]; V[3] = ta[1] - ta[3];
      }
      // (the second row's patch reads stay behind the first row's arithmetic: 32 fewer live registers)
      asm volatile("" ::: "memory");
      {
        vk_t dz[4], dw[4], tb_[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          dz[cc] = *reinterpret_cast<const vk_t*>(reinterpret_cast<const float*>(&hb_[prow[2] + cc]) + kqc);
          dw[cc] = *reinterpret_cast<const vk_t*>(reinterpret_cast<const float*>(&hb_[prow[3] + cc]) + kqc);
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) tb_[cc] = sigma * dw[cc] + dz[cc];
        V[4] = tb_[0] - tb_[2]; V[5] = tb_[1] + tb_[2]; V[6] = tb_[2] - tb_[1]; V[7] = tb_[1] - tb_[3];
      }

      // ---- 8 frequencies x NT co sub-tiles x 4 k-steps ----
      const float4* ub0 = sU + par * USL + (urow0 * KQ + kqq) * CO_T + li;
      const float4* ub1 = sU + par * USL + (urow1 * KQ + kqq) * CO_T + li;
      vk_t bf[2][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *reinterpret_cast<const vk_t*>(reinterpret_cast<const float*>(&ub0[nt * 16]) + kqc);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        if (f + 1 < 8) {
          const float4* un = (f + 1 < 4 ? ub0 : ub1) + ((f + 1) & 3) * KQ * CO_T;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            bf[(f + 1) & 1][nt] = *reinterpret_cast<const vk_t*>(reinterpret_cast<const float*>(&un[nt * 16]) + kqc);
        }
#pragma unroll
        for (int s = 0; s < KQ; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[f][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[f][s], bf[f & 1][nt][s], acc[f][nt], 0, 0, 0);
#pragma unroll
        for (int k_ = 0; k_ < PERF; ++k_)
          if (TOPP + f * PERF + k_ < NPIECE) W9_NEXT(TOPP + f * PERF + k_)
      }
#undef W9_NEXT
      if (last) {
        asm volatile("" ::: "memory");  // program order DMA -> residual loads (the vmcnt(24) above counts on it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned ro = has_res ? voff[r] : EGN_OOB;
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) rv[r][pb][nt] = wino_load4(rr, ro, pb * colpitch + nt * 64u);
        }
      }
      par ^= 1;
    }

    W9_CLK()
    // ---- output transform: t of the wave's first / second frequency row; keep = t0 + t1, send = t1 ----
    float keep[NT][4][2], send[NT][4][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t00 = acc[0][nt][r] + acc[1][nt][r] + acc[2][nt][r];
        const float t10 = acc[4][nt][r] + acc[5][nt][r] + acc[6][nt][r];
        const float t01 = acc[1][nt][r] - acc[2][nt][r] - acc[3][nt][r];
        const float t11 = acc[5][nt][r] - acc[6][nt][r] - acc[7][nt][r];
        keep[nt][r][0] = t00 + t10;
        keep[nt][r][1] = t01 + t11;
        send[nt][r][0] = t10;
        send[nt][r][1] = t11;
      }
    {
      float4* xch = sU + (par ^ 1) * USL;
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k4 = 0; k4 < 2 * NT; ++k4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = k4 * 4 + e;
          v[e] = send[idx >> 3][(idx >> 1) & 3][idx & 1];
        }
        *reinterpret_cast<f32x4*>(&xch[(wave * 2 * NT + k4) * 64 + lane]) = v;
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // fh = 0: Y0 = (t0 + t1) + t2 = keep + recv;  fh = 1: Y1 = t1 - (t2 + t3) = recv - keep = sigma * keep + recv
#pragma unroll
      for (int k4 = 0; k4 < 2 * NT; ++k4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&xch[((wave ^ MTILES) * 2 * NT + k4) * 64 + lane]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = k4 * 4 + e;
          keep[idx >> 3][(idx >> 1) & 3][idx & 1] = sigma * keep[idx >> 3][(idx >> 1) & 3][idx & 1] + v[e];
        }
      }
    }
    W9_CLK()
    float st1[NT], st2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) st1[nt] = st2[nt] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          float v = fmaxf(keep[nt][r][pb] * sc[nt] + sh[nt] + rv[r][pb][nt], act_lo);
          wino_store4(ry, voff[r], pb * colpitch + nt * 64u, v);
          if (a.stats != nullptr && voff[r] != EGN_OOB) { st1[nt] += v; st2[nt] += v * v; }
        }
    if (a.stats != nullptr) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        st1[nt] += __shfl_xor(st1[nt], 16);
        st1[nt] += __shfl_xor(st1[nt], 32);
        st2[nt] += __shfl_xor(st2[nt], 16);
        st2[nt] += __shfl_xor(st2[nt], 32);
      }
      if (kq == 0) {
        double* srow = sS + wave * 2 * CO_T + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          srow[nt * 16] += (double)st1[nt];
          srow[CO_T + nt * 16] += (double)st2[nt];
        }
      }
    }
    W9_CLK()
    tile = tile_n; ct = ct_n; tb = tb_n; ty = ty_n; tx = tx_n;
  }
  if constexpr (CLK != 0) {
    __syncthreads();
    unsigned long long* out = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res)) +
                              (size_t)blockIdx.x * (NW * W9_NTK + 1);
    for (int e = tid; e < NW * W9_NTK; e += NTH) out[1 + e] = sT[e];
    if (tid == 0) out[0] = (unsigned long long)ntk;
  }
  if (a.stats != nullptr) {
    __syncthreads();
    double* row = a.stats + (size_t)blockIdx.x * 2 * Co;
    for (int e = tid; e < 2 * Co; e += NTH) {
      const int which = e / Co, c = e - which * Co;
      const int cl = c - ct_block * CO_T;
      double v = 0.0;
      if (cl >= 0 && cl < CO_T) {
        for (int k = 0; k < NW; ++k) v += sS[(k * 2 + which) * CO_T + cl];
      }
      row[e] = v;
    }
  }
#undef W9_ITEM
#undef W9_DOFF
#undef W9_PIECE
#undef W9_CLK
}

template <int TH, int TW, int TNB, int ABL = 0>
static int wino_launch(const ConvArgs& a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];
  static int cus = 0;
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<TH, TW, TNB, ABL>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  }
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
    cus &= ~7;  // whole XCD rounds
    if (cus <= 0) cus = 8;
  }
  const int ntile = a.tiles_x * a.tiles_y * ((a.N + TNB - 1) / TNB);
  const int nwork = ((ntile + 7) / 8) * 8 * (a.Cout / WN_CO);
  const int grid = nwork < cus ? nwork : cus;
  hipLaunchKernelGGL((conv_wino_kernel<TH, TW, TNB, ABL>), dim3(grid), dim3(WN_NTH), lds, stream, a);
  return (int)hipGetLastError();
}

// persistent grid of the frequency-halves kernel: at most one block per CU, a multiple of 8 * nct so that
// (a) blocks w, w+8, ... stay on one XCD and (b) a block's items all have the same co-tile
// (ct = (w >> 3) % nct and w advances by the grid size) -- its BatchNorm partial row covers one co-tile
static int wino8_grid(const ConvArgs& a, int tnb, int blocks_per_cu = 1) {
  const int cot = egn_wino_cot(a.Cout);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int nct = a.Cout / (cot ? cot : 48);
  const int ntile = a.tiles_x * a.tiles_y * ((a.N + tnb - 1) / tnb);
  const int nwork = ((ntile + 7) / 8) * 8 * nct;
  int cap = blocks_per_cu * cus / (8 * nct) * (8 * nct);
  if (cap <= 0) cap = 8 * nct;
  return nwork < cap ? nwork : cap;
}

template <int TH, int TW, int TNB, int ABL, int NW, int NT>
static int wino8_launch_nt(const ConvArgs& a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino8_kernel<TH, TW, TNB, ABL, NW, NT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  }
  const int grid = wino8_grid(a, TNB);
  hipLaunchKernelGGL((conv_wino8_kernel<TH, TW, TNB, ABL, NW, NT>), dim3(grid), dim3(64 * NW), lds, stream, a);
  return (int)hipGetLastError();
}
template <int TH, int TW, int TNB, int ABL = 0, int NW = 8>
static int wino8_launch(const ConvArgs& a, size_t lds, hipStream_t stream) {
  if (egn_wino_cot(a.Cout) == 48) return wino8_launch_nt<TH, TW, TNB, ABL, NW, 3>(a, lds, stream);
  if constexpr (ABL == 0) {
    if (egn_wino_cot(a.Cout) == 32) return wino8_launch_nt<TH, TW, TNB, 0, NW, 2>(a, lds, stream);
  }
  return EGN_E_BADARG;
}

// ---------------------------------------------------------------------------------------------
// conv_wino43_kernel -- fused Winograd F(4x4,3x3) for the wide layers (Cout % 48 == 0, maps that are multiples
// of 16: HRNet's 96-channel 32 x 32 and 192-channel 16 x 16 branches; configs 65 / ..).  [round 3]
//
// 36 multiplies per 4x4 output patch and (ci, co) instead of the 64 of F(2x2,3x3) (144 direct): the matrix
// pipe, which bounds these layers (and on which fp32 MFMAs and every VALU instruction compete for the same FMA
// lanes, profiles/r3_mfma_tax.txt), gets 1.78x less to do.  Price: transforms with the constants of the points
// (0, +-1, +-2) -- whole-network fp32 emulation (tools/wino43_network_study.py): heat-maps within 2.5e-5 of the
// fp32 oracle (bar 5e-4), arg-max unchanged, soft-arg-max within 7e-5 px (bar 1e-3).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A,   d = 6x6 input patch, 36 frequencies f = (i, j)
//
//   * block = ONE m-tile: the 16 Winograd tiles (4 x 4 output pixels each) of a 16 x 16 pixel tile, 48 output
//     channels; SIX waves, wave i owns frequency ROW i: 6 frequencies x 3 co sub-tiles = 72 accumulators, so
//     two blocks (12 waves, three per SIMD) share a CU with 2 x 70 KB of LDS;
//   * K step = 4 input channels = the four k lanes of v_mfma_f32_16x16x4_f32 (lane (tile = l & 15, kq = l >> 4)
//     holds channel kq): everything a lane touches is one dword, the halo is pixel major [slot][4] with rows
//     skewed (slot = 24 y + x + (y >> 2)) so that the 16 tiles spread over all banks (2-way conflicts, the
//     minimum for 16 tiles x 2 channels on 32 banks);
//   * input transform per wave: T[i][c] = sum_a B^T[i][a] d[a][c] -- row i of B^T has at most four non-zeros,
//     one of them 1: three v_fma with wave-uniform coefficients, which rows a are read is a per-wave LDS
//     offset -- then the 6-point transform along the row (12 operations): 30 VALU per 18 MFMAs;
//   * U = G g G^T transformed in float64 on the host and rounded once, packed [co-tile][K step][f][kq][48];
//   * output transform: wave i forms t_i[b] = sum_j M[i][j] A[j][b] (10 operations per accumulator vector), the
//     six waves exchange them through LDS one co sub-tile at a time, waves 0..3 finish the output rows
//     a = 0..3 of every tile: Y[a][b] = sum_i A^T[a][i] t_i[b], scale / shift / residual / ReLU, dword stores.
// Persistent blocks over (tile, co-tile) items, XCD-aware like conv_wino8_kernel; the next item's first stage is
// prefetched during the last K step.  Stage ring of two, ONE barrier per K step.
namespace {
constexpr int W43_RP = 24;                       // halo row pitch in pixel slots
constexpr int W43_HS = 448;                      // halo slots per m-tile and stage (17 * 24 + 17 + 4 < 448 = 7 x 64)
constexpr int W43_HP = W43_HS / 64;              // 7 DMA pieces per m-tile
constexpr int W43_MT = 2;                        // m-tiles (16 x 16 pixel tiles) per block
constexpr int W43_UF = 36 * 4 * 48;              // floats per (co-tile, K step) slab of U: 27 648 B
constexpr int W43_UP = W43_UF / 256;             // 27 DMA pieces
constexpr int W43_NP = W43_MT * W43_HP + W43_UP; // 41 pieces per stage
constexpr int W43_HF = W43_HS * 4;               // floats of one halo image
constexpr int W43_STAGE = W43_MT * W43_HF + W43_UF;   // floats per stage: 10 496 (41 984 B)
constexpr int W43_RING = 3;                      // stages: the DMA of a K step has two K steps to land
constexpr int W43_NW = 6 * W43_MT;               // 12 waves: (m-tile, frequency row)
constexpr int W43_NTH = 64 * W43_NW;
constexpr int W43_KP = (W43_NP + W43_NW - 1) / W43_NW;   // 4 piece slots per wave; the last one only for waves < 5
}  // namespace

// K step = 4 input channels; a wave's DMA pieces of a stage: p = k * 12 + wave (k = 0..3), p < 14: halo of m-tile
// p / 7, else U slab.  Ring of three stages: the pieces of K step g + 2 go out during step g (into the stage step
// g - 1 consumed), so every piece has two whole steps to land -- with 18 MFMAs per wave and step a two-stage ring
// made every step wait a full DMA round trip (first version: 69 us where this one takes 3x.. see DESIGN).
template <int CLK = 0>
__global__ __launch_bounds__(W43_NTH, 3) void conv_wino43_kernel(ConvArgs a) {
  extern __shared__ float4 smem[];
  float* sm = reinterpret_cast<float*>(smem);
  const unsigned lds0 = wino_lds_addr(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave / 6;                         // m-tile of the block
  const int fr = wave - 6 * mt;                    // frequency row i
  const int li = lane & 15, kq = lane >> 4;

  const int C = a.Cin, Co = a.Cout;
  const int nct = Co / 48;
  const int nsteps = C / 4;
  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const unsigned long long uaddr = reinterpret_cast<unsigned long long>(a.w);
  const u32x4 rxv = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, (unsigned)((size_t)a.N * a.H * a.W * C * 4), 0x00020000u};
  const u32x4 ruv = {(unsigned)uaddr, (unsigned)(uaddr >> 32) & 0xffffu, (unsigned)((size_t)nct * nsteps * W43_UF * 4), 0x00020000u};
  const unsigned out_bytes = (unsigned)((size_t)a.N * a.Ho * a.Wo * Co * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, out_bytes, 0x00020000);

  // ---- halo pieces of this wave: k = 0 -> p = wave (all 12 are halo pieces), k = 1 -> p = 12 + wave (waves 0, 1) ----
  int hyx[2], hrel[2], hmt[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = k * W43_NW + wave;
    const int m_ = p / W43_HP, q = p - m_ * W43_HP;
    const int e = q * 64 + lane;
    const int y = e / W43_RP, x = e - y * W43_RP - (y >> 2);
    const bool ok = p < W43_MT * W43_HP && y < 18 && x >= 0 && x < 18;
    hyx[k] = ok ? ((y << 8) | x) : -1;
    hrel[k] = ok ? (y * a.W + x) * C * 4 : 0;
    hmt[k] = m_ < W43_MT ? m_ : 0;
  }
  // ---- B^T row of this wave: T = c0 d[r0] + c1 d[r1] + c2 d[r2] + d[r3] ----
  //   row 0: 4 d0 - 5 d2 + d4      1: -4 d1 - 4 d2 + d3 + d4     2: 4 d1 - 4 d2 - d3 + d4
  //       3: -2 d1 - d2 + 2 d3 + d4    4: 2 d1 - d2 - 2 d3 + d4      5: 4 d1 - 5 d3 + d5
  int r0, r1, r2, r3;
  float c0, c1, c2;
  switch (fr) {
    case 0: r0 = 0; r1 = 2; r2 = 2; r3 = 4; c0 = 4.f; c1 = -5.f; c2 = 0.f; break;
    case 1: r0 = 1; r1 = 2; r2 = 3; r3 = 4; c0 = -4.f; c1 = -4.f; c2 = 1.f; break;
    case 2: r0 = 1; r1 = 2; r2 = 3; r3 = 4; c0 = 4.f; c1 = -4.f; c2 = -1.f; break;
    case 3: r0 = 1; r1 = 2; r2 = 3; r3 = 4; c0 = -2.f; c1 = -1.f; c2 = 2.f; break;
    case 4: r0 = 1; r1 = 2; r2 = 3; r3 = 4; c0 = 2.f; c1 = -1.f; c2 = -2.f; break;
    default: r0 = 1; r1 = 3; r2 = 3; r3 = 5; c0 = 4.f; c1 = -5.f; c2 = 0.f; break;
  }
  // float index (inside a stage) of patch element (row a, column 0) of this lane's tile (ty = li >> 2, tx = li & 3)
  const int tyl = li >> 2, txl = li & 3;
#define W43_PROW(A) (mt * W43_HF + (((4 * tyl + (A)) * W43_RP + 4 * txl + tyl + ((A) >> 2)) * 4) + kq)
  const int pr0 = W43_PROW(r0), pr1 = W43_PROW(r1), pr2 = W43_PROW(r2), pr3 = W43_PROW(r3);
#undef W43_PROW
  // float index of this lane's B fragment of frequency (fr, 0), co sub-tile 0
  const int ub = W43_MT * W43_HF + ((fr * 6) * 4 + kq) * 48 + li;

  const int tiles_xy = a.tiles_x * a.tiles_y;
  const int ntile = tiles_xy * a.N;
  const int npair = (ntile + 1) >> 1;              // items = pairs of consecutive tiles (a row of a 32-wide map, or two images)
  const int nwork = ((npair + 7) >> 3) * nct * 8;
  const int gsz = __builtin_amdgcn_readfirstlane((int)gridDim.x);

  unsigned long long* sT = nullptr;
  int ntk = 0;
  if constexpr (CLK != 0) sT = reinterpret_cast<unsigned long long*>(sm + W43_RING * W43_STAGE);
#define W43_CLK()                                                                        \
  {                                                                                      \
    if constexpr (CLK != 0) {                                                            \
      if (lane == 0 && ntk < 48) sT[wave * 48 + ntk] = __builtin_readcyclecounter();     \
      ++ntk;                                                                             \
    }                                                                                    \
  }

// work index -> (tile pair, co-tile); TILE (0 / 1) = 2 * pair + m-tile -> (n, ty, tx), all on the scalar unit
#define W43_ITEM(Wi, PAIR_, CT_)                                                 \
  {                                                                              \
    const unsigned wi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(Wi));    \
    const unsigned x_ = wi_ & 7u, q_ = wi_ >> 3;                                 \
    const unsigned qq_ = wn_udiv(q_, a.mg_nct);                                  \
    PAIR_ = (int)(qq_ * 8u + x_);                                                \
    CT_ = (int)(q_ - qq_ * (unsigned)nct);                                       \
  }
#define W43_TILE(TILE_, N_, TY_, TX_)                                            \
  {                                                                              \
    const unsigned tbq_ = wn_udiv((unsigned)(TILE_), a.mg_txy);                  \
    const unsigned rq_ = (unsigned)(TILE_)-tbq_ * (unsigned)tiles_xy;            \
    const unsigned tyq_ = wn_udiv(rq_, a.mg_tx);                                 \
    N_ = (int)tbq_; TY_ = (int)tyq_; TX_ = (int)(rq_ - tyq_ * (unsigned)a.tiles_x); \
  }
// halo DMA byte offsets (K step 0) of this wave's halo pieces for tile pair PAIR_
#define W43_DOFF(PAIR_, OUT)                                                                                 \
  {                                                                                                          \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                          \
      const int t_ = 2 * (PAIR_) + hmt[k];                                                                   \
      int n_, ty_2, tx_2;                                                                                    \
      W43_TILE(t_, n_, ty_2, tx_2)                                                                           \
      const int iy0_ = ty_2 * 16 - 1, ix0_ = tx_2 * 16 - 1;                                                  \
      const int base_ = ((n_ * a.H + iy0_) * a.W + ix0_) * C * 4;                                            \
      const unsigned iy_ = (unsigned)(iy0_ + (hyx[k] >> 8)), ix_ = (unsigned)(ix0_ + (hyx[k] & 255));        \
      const bool in_ = t_ < ntile && hyx[k] >= 0 && iy_ < (unsigned)a.H && ix_ < (unsigned)a.W;              \
      OUT[k] = in_ ? (unsigned)(base_ + hrel[k]) : EGN_OOB;                                                  \
    }                                                                                                        \
  }
// DMA piece slot K (0..3) of this wave for ring stage P: channel group STEP, U slab (CT, STEP)
#define W43_PIECE(K, P, OFF, CT, STEP)                                                                        \
  {                                                                                                          \
    const int p_ = (K)*W43_NW + wave;                                                                        \
    if (p_ < W43_MT * W43_HP) {                                                                              \
      wino_dma16(rxv, lds0 + (unsigned)((P)*W43_STAGE * 4 + p_ * 1024), OFF[(K) < 2 ? (K) : 0], (unsigned)(STEP)*16u); \
    } else if (p_ < W43_NP) {                                                                                \
      wino_dma16(ruv, lds0 + (unsigned)((P)*W43_STAGE * 4 + W43_MT * W43_HF * 4 + (p_ - W43_MT * W43_HP) * 1024), \
                 (unsigned)lane * 16u, (unsigned)((((CT)*nsteps + (STEP)) * W43_UP + (p_ - W43_MT * W43_HP)) * 1024)); \
    }                                                                                                        \
  }
#define W43_ISSUE(P, OFF, CT, STEP) { _Pragma("unroll") for (int k_ = 0; k_ < W43_KP; ++k_) W43_PIECE(k_, P, OFF, CT, STEP) }

  int w = blockIdx.x;
  int pair = 0, ct = 0;
  W43_ITEM(w, pair, ct)
  unsigned doff[2];
  W43_DOFF(pair, doff)
  int rs = 0;                                    // ring stage of the K step about to be computed
  if (w < nwork) {
    W43_ISSUE(0, doff, ct, 0)
    if (nsteps > 1) W43_ISSUE(1, doff, ct, 1)
  }
  bool first = true;
  const float act_lo = (a.act & EGN_ACT_MASK) == EGN_ACT_RELU ? 0.f : -__builtin_inff();
  const bool has_res = CLK ? false : a.res != nullptr;
  const unsigned colpitch = (unsigned)Co * 4u;
  const bool fin = fr < 4;                       // waves with fr = 0..3 finish output row a = fr of every tile of their m-tile
  const bool pw4 = wave < W43_NP - (W43_KP - 1) * W43_NW;   // this wave has a piece in the last slot (4 pieces per stage, else 3)
  W43_CLK()

  for (; w < nwork; w += gsz) {
    int pair_n = 0, ct_n = 0;
    W43_ITEM(w + gsz, pair_n, ct_n)
    const bool more = (w + gsz) < nwork;
    unsigned doff_n[2];
    W43_DOFF(pair_n, doff_n)
    // this wave's m-tile: output pixel (4 kq + a, 4 r + b) of tile row kq: byte offset of (row a = fr, column 0)
    unsigned vbase;
    {
      const int t_ = 2 * pair + mt;
      int n_, ty_, tx_;
      W43_TILE(t_, n_, ty_, tx_)
      vbase = (t_ < ntile) ? (unsigned)((((n_ * a.Ho + ty_ * 16 + 4 * kq + (fin ? fr : 0)) * a.Wo + tx_ * 16) * Co + ct * 48 + li) * 4)
                           : EGN_OOB;
    }
    float sc[3], sh[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      sc[nt] = a.scale[ct * 48 + nt * 16 + li];
      sh[nt] = a.shift[ct * 48 + nt * 16 + li];
    }
    f32x4 acc[6][3];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nsteps; ++c) {
      asm volatile("" ::: "memory");
      W43_CLK()
      // The stage of this K step has landed in this wave's share.  In flight behind it (issue order): the next
      // K step's pieces (3 or 4 of this wave) -- except at the last step of an item, whose successor was not
      // issued yet -- and, at the first step after an epilogue, the 48 stores of a finishing wave.
      {
        const bool tail_free = (c + 1 == nsteps);            // nothing newer than this stage (but the stores)
        if (c == 0 && !first && fin) {
          // [stage of this step] [48 stores]: the stage was issued before the epilogue
          __builtin_amdgcn_s_waitcnt(0xC070);                                // vmcnt(48)
        } else if (tail_free || (c == 0 && !first)) {
          __builtin_amdgcn_s_waitcnt(0x0070);                                // vmcnt(0)
        } else if (pw4) {
          __builtin_amdgcn_s_waitcnt(0x0074);                                // vmcnt(4)
        } else {
          __builtin_amdgcn_s_waitcnt(0x0073);                                // vmcnt(3)
        }
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      W43_CLK()
      // ---- issue: K step c + 2 of this item into the stage step c - 1 consumed; the item's first step also issues
      // its step 1 (deferred: at the end of the previous item only stage 0 of this one was prefetched, the other
      // two stages served the exchange); the item's LAST-BUT-ONE step prefetches the next item's step 0 ----
      int st2 = rs + 2; if (st2 >= W43_RING) st2 -= W43_RING;
      int st1 = rs + 1; if (st1 >= W43_RING) st1 -= W43_RING;
#define W43_NEXT(K)                                                               \
  {                                                                               \
    __builtin_amdgcn_sched_barrier(0x0106);                                       \
    if (c + 2 < nsteps) W43_PIECE(K, st2, doff, ct, c + 2)                        \
    else if (c + 2 == nsteps && more) W43_PIECE(K, st2, doff_n, ct_n, 0)          \
    __builtin_amdgcn_sched_barrier(0x0106);                                       \
  }
      if (c == 0 && !first && nsteps > 1) W43_ISSUE(st1, doff, ct, 1)
      first = false;
      const float* hb = sm + rs * W43_STAGE;
      // ---- input transform: row `fr` of B^T d, then the 6-point transform along it ----
      float t[6];
#pragma unroll
      for (int cc = 0; cc < 6; ++cc)
        t[cc] = __builtin_fmaf(c0, hb[pr0 + cc * 4], __builtin_fmaf(c1, hb[pr1 + cc * 4], __builtin_fmaf(c2, hb[pr2 + cc * 4], hb[pr3 + cc * 4])));
      float V[6];
      {
        const float u = __builtin_fmaf(-4.f, t[2], t[4]), v = __builtin_fmaf(-4.f, t[1], t[3]);
        const float p = t[4] - t[2], q = t[3] - t[1];
        V[0] = __builtin_fmaf(4.f, t[0], __builtin_fmaf(-5.f, t[2], t[4]));
        V[1] = u + v;
        V[2] = u - v;
        V[3] = __builtin_fmaf(2.f, q, p);
        V[4] = __builtin_fmaf(-2.f, q, p);
        V[5] = __builtin_fmaf(4.f, t[1], __builtin_fmaf(-5.f, t[3], t[5]));
      }
      // ---- 6 frequencies x 3 co sub-tiles, one MFMA (k = the 4 channels) each ----
      const float* ubp = hb + ub;
      float bf[2][3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) bf[0][nt] = ubp[nt * 16];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (j + 1 < 6) {
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) bf[(j + 1) & 1][nt] = ubp[(j + 1) * 192 + nt * 16];
        }
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
          acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[j], bf[j & 1][nt], acc[j][nt], 0, 0, 0);
        if (j < W43_KP) W43_NEXT(j)
      }
#undef W43_NEXT
      if (++rs == W43_RING) rs = 0;
    }
    W43_CLK()

    // ---- output transform, first half: t_i[b] = sum_j M[i][j] A[j][b]  (vectors over the 4 tile columns r) ----
    f32x4 tb[3][4];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const f32x4 p = acc[1][nt] + acc[2][nt], q = acc[1][nt] - acc[2][nt];
      const f32x4 r_ = acc[3][nt] + acc[4][nt], s_ = acc[3][nt] - acc[4][nt];
      tb[nt][0] = acc[0][nt] + p + r_;
      tb[nt][1] = q + 2.f * s_;
      tb[nt][2] = p + 4.f * r_;
      tb[nt][3] = q + 8.f * s_ + acc[5][nt];
    }
    // ---- exchange, one co sub-tile per round, through the stages that hold nothing: `rs` is the stage of the next
    // item's step 0 (prefetched at this item's last-but-one step); the stage before it (consumed by the last
    // step) and the one after it (its filling was deferred) are free: 2 x 42 KB for 12 waves x 4 KB ----
    int xs = rs + 1; if (xs >= W43_RING) xs -= W43_RING;
    // (with a ring of 3 the stage after `rs` and the stage before `rs` are (rs + 1) and (rs + 2) mod 3: m-tile 0
    //  exchanges in one, m-tile 1 in the other)
    int xs2 = rs + 2; if (xs2 >= W43_RING) xs2 -= W43_RING;
    f32x4* xch = reinterpret_cast<f32x4*>(sm + (mt == 0 ? xs : xs2) * W43_STAGE);
    float rv[2][4][4];     // residual of round nt (slot nt & 1): [b][r]
    if (fin) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[0][b][r] = wino_load4(rr, has_res ? vbase : EGN_OOB, (unsigned)(4 * r + b) * colpitch);
    }
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): own LDS reads of the K loop / of the previous round are done
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int b = 0; b < 4; ++b) xch[(fr * 4 + b) * 64 + lane] = tb[nt][b];
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (fin) {
        if (nt + 1 < 3) {      // next round's residual, one round ahead
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              rv[(nt + 1) & 1][b][r] = wino_load4(rr, has_res ? vbase : EGN_OOB, (unsigned)(4 * r + b) * colpitch + (nt + 1) * 64u);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          f32x4 ti[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) ti[i] = xch[(i * 4 + b) * 64 + lane];
          // A^T rows: (1,1,1,1,1,0) (0,1,-1,2,-2,0) (0,1,1,4,4,0) (0,1,-1,8,-8,1)
          f32x4 y;
          if (fr == 0) y = ti[0] + ti[1] + ti[2] + ti[3] + ti[4];
          else if (fr == 1) y = (ti[1] - ti[2]) + 2.f * (ti[3] - ti[4]);
          else if (fr == 2) y = (ti[1] + ti[2]) + 4.f * (ti[3] + ti[4]);
          else y = (ti[1] - ti[2]) + 8.f * (ti[3] - ti[4]) + ti[5];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = fmaxf(y[r] * sc[nt] + sh[nt] + rv[nt & 1][b][r], act_lo);
            wino_store4(ry, vbase, (unsigned)(4 * r + b) * colpitch + nt * 64u, v);
          }
        }
      }
    }
    W43_CLK()
    pair = pair_n; ct = ct_n;
    doff[0] = doff_n[0]; doff[1] = doff_n[1];
  }
  if constexpr (CLK != 0) {
    __syncthreads();
    unsigned long long* out = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res)) + (size_t)blockIdx.x * (W43_NW * 48 + 1);
    for (int e = tid; e < W43_NW * 48; e += W43_NTH) out[1 + e] = sT[e];
    if (tid == 0) out[0] = (unsigned long long)ntk;
  }
#undef W43_ITEM
#undef W43_TILE
#undef W43_DOFF
#undef W43_PIECE
#undef W43_ISSUE
#undef W43_CLK
}

size_t egn_conv_wino43_lds_bytes(int clk) { return (size_t)W43_RING * W43_STAGE * 4 + (clk ? W43_NW * 48 * 8 : 0); }

static unsigned wino_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

template <int TH, int TW, int TNB, int NW, int NT, int CLK = 0, int KQ = EGN_CKQ>
static int wino9_launch_nt(ConvArgs a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino9_kernel<TH, TW, TNB, NW, NT, CLK, KQ>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  }
  // the kernel divides its item / tile indices by nct, tiles_x * tiles_y and tiles_x with ONE multiply-high by
  // ceil(2^32 / d): exact while dividend * d < 2^32.  Dividends stay below (tiles + 8) * nct; shapes beyond that bound
  // (far outside the 2 GB tensor guards for any HRNet map, but a legal call) are refused, not mis-decoded
  {
    const unsigned long long nct_ = (unsigned long long)(a.Cout / (16 * NT));
    const unsigned long long ntile_ = (unsigned long long)a.tiles_x * a.tiles_y * ((a.N + TNB - 1) / TNB) + 8;
    if (ntile_ * nct_ * nct_ >= 0x100000000ull || ntile_ * (unsigned long long)(a.tiles_x * a.tiles_y) >= 0x100000000ull)
      return EGN_E_BADARG;
  }
  a.mg_nct = wino_magic(a.Cout / (16 * NT));
  a.mg_txy = wino_magic(a.tiles_x * a.tiles_y);
  a.mg_tx = wino_magic(a.tiles_x);
  const int grid = wino8_grid(a, TNB, KQ == EGN_CKQ ? 1 : 2);     // half-chunk stages: two blocks per CU
  hipLaunchKernelGGL((conv_wino9_kernel<TH, TW, TNB, NW, NT, CLK, KQ>), dim3(grid), dim3(64 * NW), lds, stream, a);
  return (int)hipGetLastError();
}
template <int TH, int TW, int TNB, int NW = 8, int KQ = EGN_CKQ>
static int wino9_launch(const ConvArgs& a, size_t lds, hipStream_t stream) {
  if (egn_wino_cot(a.Cout) == 48) return wino9_launch_nt<TH, TW, TNB, NW, 3, 0, KQ>(a, lds, stream);
  if (egn_wino_cot(a.Cout) == 32) return wino9_launch_nt<TH, TW, TNB, NW, 2, 0, KQ>(a, lds, stream);
  return EGN_E_BADARG;
}

template <int CLK>
static int wino43_launch(ConvArgs a, size_t lds, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];
  static int cus = 0;
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino43_kernel<CLK>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  }
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int nct = a.Cout / 48;
  a.mg_nct = wino_magic(nct);
  a.mg_txy = wino_magic(a.tiles_x * a.tiles_y);
  a.mg_tx = wino_magic(a.tiles_x);
  const int ntile = a.tiles_x * a.tiles_y * a.N;
  const int npair = (ntile + 1) / 2;
  const int nwork = ((npair + 7) / 8) * 8 * nct;
  int cap = cus / (8 * nct) * (8 * nct);              // one 126 KB block per CU, whole XCD rounds
  if (cap <= 0) cap = 8 * nct;
  const int grid = nwork < cap ? nwork : cap;
  hipLaunchKernelGGL((conv_wino43_kernel<CLK>), dim3(grid), dim3(W43_NTH), lds, stream, a);
  return (int)hipGetLastError();
}

// rows of the BatchNorm partial table a launch writes (0 = this variant has no fused statistics)
int egn_conv_wino_stats_rows(const ConvArgs& a, int variant) {
  int v = variant & 15;
  if ((variant >> 4) || v < 2 || v == 10 || v > 12) return 0;      // (10: the F(4x4,3x3) kernel has no fused statistics)
  const int per_cu = v >= 11 ? 2 : 1;
  if (v >= 11) v = v == 11 ? 5 : 4;   // 11 / 12: conv_wino9_kernel with 8-channel stages on the tiles of 5 / 4
  if (v >= 6) v -= 4;              // variants 6..9 = conv_wino9_kernel on the geometries of 2..5
  const int tnb = v == 3 ? 4 : (v == 4 ? 2 : 1);
  return wino8_grid(a, tnb, per_cu);       // one partial row per block
}

// variant 0: 16 x 16 pixel tile of one image; variant 1: four 8 x 8 images (the 8 x 8 maps);
// variants 2 / 3: the same two geometries on the 8-wave kernel; variant 4: two 8 x 8 images, 4 waves
size_t egn_conv_wino_lds_bytes(int variant, int cout) {
  if ((variant & 15) == 10) return egn_conv_wino43_lds_bytes(variant >> 4);
  int v = variant & 15;
  if (v == 11 || v == 12) {        // 8-channel stages, 4 waves: half the filter stage and half the halo planes
    const int cot = egn_wino_cot(cout) ? egn_wino_cot(cout) : WN_CO;
    const size_t halo2 = 2 * (size_t)(v == 11 ? WinoGeom<8, 16, 1>::PLANE : WinoGeom<8, 8, 2>::PLANE);
    return (2 * (size_t)(16 * 2 * cot) + 2 * halo2) * 16 + (size_t)4 * 2 * cot * sizeof(double) +
           ((variant >> 4) == 4 ? 4 * 48 * sizeof(unsigned long long) : 0);     // stamp build
  }
  if (v >= 6) v -= 4;              // conv_wino9_kernel: the LDS image of conv_wino8_kernel
  size_t halo = (v & 1) ? WinoDims<8, 8, 4>::BUF : WinoDims<16, 16, 1>::BUF;
  if (v >= 2) halo = (v & 1) ? EGN_CKQ * WinoGeom<8, 8, 4>::PLANE : EGN_CKQ * WinoGeom<16, 16, 1>::PLANE;
  if (v == 4) halo = EGN_CKQ * WinoGeom<8, 8, 2>::PLANE;
  if (v == 5) halo = EGN_CKQ * WinoGeom<8, 16, 1>::PLANE;
  const int cot = v >= 2 && egn_wino_cot(cout) ? egn_wino_cot(cout) : WN_CO;   // the 4-wave kernel: 48 only
  const size_t stats = v >= 2 ? (size_t)((v == 4 || v == 5) ? 4 : 8) * 2 * cot * sizeof(double) : 0;
  const size_t stamps = (variant >> 4) == 4 && v >= 2 ? 8 * 48 * sizeof(unsigned long long) : 0;   // stamp builds
  return (2 * (size_t)(16 * EGN_CKQ * cot) + 2 * halo) * 16 + stats + stamps;
}
int egn_conv_launch_wino(const ConvArgs& a, size_t lds, int variant, hipStream_t stream) {
  const int act = a.act & EGN_ACT_MASK;
  if ((act != EGN_ACT_NONE && act != EGN_ACT_RELU) || (a.act & EGN_ACT_RES_AFTER)) return EGN_E_BADARG;
  switch (variant) {
    // what the shipped table / the tuner can select (egn_conv_config_kind >= 0 in the product build)
    case 2: return wino8_launch<16, 16, 1>(a, lds, stream);
    case 3: return wino8_launch<8, 8, 4>(a, lds, stream);
    case 4: return wino8_launch<8, 8, 2, 0, 4>(a, lds, stream);
    case 5: return wino8_launch<8, 16, 1, 0, 4>(a, lds, stream);
    case 6: return wino9_launch<16, 16, 1>(a, lds, stream);
    case 7: return wino9_launch<8, 8, 4>(a, lds, stream);
    case 8: return wino9_launch<8, 8, 2, 4>(a, lds, stream);
    case 9: return wino9_launch<8, 16, 1, 4>(a, lds, stream);
#ifdef EGN_PROBES
    // -DEGN_PROBES (python -m egonet_amd.build --probes; tools/ only): the families measured and retired -- the
    // 4-wave kernel (cfg 45 / 46), the first F(4x4,3x3) kernel (65), two 4-wave blocks per CU (67 / 68) -- and the
    // timing-ablation / s_memtime-stamp builds (WRONG RESULTS; 47-50, 53-55, 58, 63, 66, 69)
    case 0: return wino_launch<16, 16, 1>(a, lds, stream);
    case 1: return wino_launch<8, 8, 4>(a, lds, stream);
    case 10: return a.stats ? EGN_E_BADARG : wino43_launch<0>(a, lds, stream);
    case 0x1a: return wino43_launch<1>(a, lds, stream);           // s_memtime stamps (tools/wino_clk.py)
    case 11: return wino9_launch<8, 16, 1, 4, 2>(a, lds, stream);   // 8-channel stages, two blocks per CU
    case 12: return wino9_launch<8, 8, 2, 4, 2>(a, lds, stream);
    case 0x12: return wino8_launch<16, 16, 1, 16>(a, lds, stream);
    case 0x22: return wino8_launch<16, 16, 1, 7>(a, lds, stream);
    case 0x32: return wino8_launch<16, 16, 1, 3>(a, lds, stream);
    case 0x42: return wino8_launch<16, 16, 1, 32>(a, lds, stream);   // timeline stamps (tools/wino_clk.py)
    case 0x46: return egn_wino_cot(a.Cout) == 48 ? wino9_launch_nt<16, 16, 1, 8, 3, 1>(a, lds, stream) : EGN_E_BADARG;
    case 0x4b: return egn_wino_cot(a.Cout) == 48 ? wino9_launch_nt<8, 16, 1, 4, 3, 1, 2>(a, lds, stream) : EGN_E_BADARG;
    case 0x10: return wino_launch<16, 16, 1, 15>(a, lds, stream);
    case 0x20: return wino_launch<16, 16, 1, 7>(a, lds, stream);
    case 0x30: return wino_launch<16, 16, 1, 3>(a, lds, stream);
    case 0x40: return wino_launch<16, 16, 1, 11>(a, lds, stream);
#endif
    default: return EGN_E_BADARG;
  }
}

// ---------------------------------------------------------------------------------------------
// Filter transform on the device: torch weight [Cout][Cin][3][3] -> U = G g G^T in the packed layout
// above ([co-tile][chunk][f][quad][48][4]).  float64 arithmetic, one rounding to fp32.  dgrad = 1 packs
// the data-gradient filter (in/out channels swapped, taps rotated by 180 degrees) like
// egn_pack_conv_weight_f32.  One thread per (co, ci): 9 loads, 16 stores.
__global__ __launch_bounds__(256) void wino_pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                               int dgrad, float* __restrict__ dst) {
  const int n_out = dgrad ? Cin : Cout, n_in = dgrad ? Cout : Cin;
  const int nchunk = n_in / EGN_CK;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_out * n_in; e += gridDim.x * blockDim.x) {
    const int o = e / n_in, i = e - o * n_in;
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        g[a][b] = dgrad ? (double)w[((size_t)i * Cin + o) * 9 + (2 - a) * 3 + (2 - b)]
                        : (double)w[((size_t)o * Cin + i) * 9 + a * 3 + b];
    double t[4][3];  // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
    const int cot = egn_wino_cot(n_out);
    const int ct = o / cot, col = o - ct * cot;
    const int chunk = i / EGN_CK, q = (i % EGN_CK) >> 2, r = i & 3;
    float* base = dst + ((size_t)(ct * nchunk + chunk) * (16 * EGN_CKQ * cot)) * 4 + ((size_t)q * cot + col) * 4 + r;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double u0 = t[a][0], u1 = 0.5 * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5 * (t[a][0] - t[a][1] + t[a][2]),
                   u3 = t[a][2];
      const double u[4] = {u0, u1, u2, u3};
#pragma unroll
      for (int b = 0; b < 4; ++b) base[(size_t)(a * 4 + b) * EGN_CKQ * cot * 4] = (float)u[b];
    }
  }
}

extern "C" long egn_wino_weight_floats(int Cout, int Cin, int dgrad) {
  const int n_out = dgrad ? Cin : Cout, n_in = dgrad ? Cout : Cin;
  if (n_out <= 0 || n_in <= 0 || egn_wino_cot(n_out) == 0 || n_in % EGN_CK) return 0;
  return (long)n_out * n_in * 16;
}
extern "C" int egn_wino_pack_weight_f32(const float* w, int Cout, int Cin, int dgrad, float* dst, void* stream) {
  if (!w || !dst || egn_wino_weight_floats(Cout, Cin, dgrad) == 0) return EGN_E_BADARG;
  const long total = (long)Cout * Cin;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(wino_pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, dgrad, dst);
  return (int)hipGetLastError();
}
