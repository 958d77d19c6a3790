// conv_c48.hip -- the 48 -> 48 channel 3x3 convolution (stride 1, pad 1, NHWC) of
// HRNet's full-resolution branch with the WHOLE FILTER RESIDENT IN LDS.
//
// Why a dedicated kernel: these layers (hrnet.py:76-92 BasicBlock convs of branch 0,
// 89 of the 306 conv launches of a W48 forward, 36 % of its time) have the shortest
// K loop of the network (48 x 9 = 432), so what the general kernels pay per tile --
// a weight slab DMA and a barrier per K stage, a prologue that cannot overlap
// anything, a store burst at the end of every block -- is the largest fraction
// there (62 % of the fp32-MFMA peak vs 70+ % on the wider branches).  The filter is
// only 48*48*9*4 B = 82.9 KB in the packed layout, i.e. it FITS in the 160 KB LDS
// next to a double-buffered 48-channel halo tile (2 x 36.9 KB).  So:
//
//   * one persistent block per CU loads the filter once and then walks over pixel
//     tiles (8 x 16 outputs) with a stride of gridDim.x;
//   * per tile: ONE barrier; the next tile's halo (all three 16-channel chunks)
//     is LDS-DMA'd into the other buffer while the 27 (chunk, tap) steps of the
//     current one run straight through -- no barrier, no DMA issue, no waitcnt on
//     global memory inside the K loop;
//   * the epilogue is wave-private and straight from the accumulators: a lane of a
//     16x16 MFMA tile owns one channel of 4 consecutive pixels, 16 lanes cover 64
//     contiguous bytes of a pixel's 192-byte row, so scale/shift, residual, ReLU
//     and the dword buffer stores need no LDS pass and no barrier; the stores are
//     not waited for (s_waitcnt vmcnt(24) before the next barrier leaves them in
//     flight), they overlap the next tile's MFMAs.
//
// Fragment mapping, weight packing and numerics are those of conv_dma.hip /
// conv_mfma.hip (v_mfma_f32_16x16x4_f32, exact fp32); 4 waves, wave w owns tile
// rows 2w and 2w+1 (MT = 2) x all 48 output channels (NT = 3).
#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_c48_t;

__device__ __forceinline__ void c48_dma16(__amdgpu_buffer_rsrc_t r, float4* dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_c48_t)dst, 16, voff, 0, 0, 0);
}
// The per-tile halo DMA as raw ISA.  Through the builtin the compiler knows that the
// instruction writes LDS and puts an s_waitcnt vmcnt(0) in front of the next ds_read --
// of the OTHER buffer -- which would expose the latency of the very prefetch this kernel
// exists for.  The asm is invisible to the waitcnt insertion pass; completion is
// enforced by hand (s_waitcnt vmcnt(24) + s_barrier at the top of the tile loop).
// lds_addr: wave-uniform LDS byte address of the wave's 64 x 16 B destination.
__device__ __forceinline__ void c48_dma16_raw(u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc)
               : "m0");
}
__device__ __forceinline__ unsigned c48_lds_addr(const void* p) {
  return (unsigned)(__UINTPTR_TYPE__)(lds_ptr_c48_t) const_cast<void*>(p);  // LDS pointers are 32-bit offsets
}
__device__ __forceinline__ float c48_load4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ void c48_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, 0, 0);
}

namespace {
constexpr int C48 = 48, TH = 8, TW = 16, HH = TH + 2, HWD = TW + 2;
constexpr int NPIXP = 192;                       // 180 halo pixels rounded up to 16
constexpr int CHUNK_SLOTS = NPIXP * EGN_CKQ;     // float4 per 16-channel chunk of a halo buffer
constexpr int HALO_SLOTS = 3 * CHUNK_SLOTS;      // 2304 float4 = 36.9 KB
constexpr int W_SLOTS = 3 * 9 * EGN_CKQ * C48;   // 5184 float4 = 82.9 KB
constexpr int NT = 3;
}  // namespace

// WAVES = 4: wave w owns tile rows 2w, 2w+1 (MT = 2); WAVES = 8: one row each (MT = 1), two
// waves per SIMD so that one wave's epilogue / barrier wait hides under the other's MFMAs
//
// REGF (4 waves only): every wave copies ALL its B fragments (27 steps x 3 float4 = 324 registers,
// one wave per SIMD owns the whole 512-entry file) out of LDS once, and the K loop reads only the
// A fragments: 24 MFMAs per 2 ds_read_b128 instead of 12 per 4.  In isolation
// (tools/micro/mfma_lds.hip vs mfma_regfilter.hip) that loop runs at 141 TFLOP/s where the
// LDS-fed one reaches 113-125: the ds_reads woven between the MFMAs are what the pipe waits for.
template <int WAVES, bool REGF>
__device__ __forceinline__ void c48_body(const ConvArgs& a) {
  constexpr int NTH = 64 * WAVES;
  constexpr int MT = TH / WAVES;
  constexpr int A_IT = (HALO_SLOTS + NTH - 1) / NTH;  // DMA instructions per lane and tile
  extern __shared__ float4 smem[];
  float4* sW = smem;             // [chunk][tap][quad][48]   (the packed filter, verbatim)
  float4* sA = smem + W_SLOTS;   // [2][chunk][pixel][quad]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;
  const int kq = lane >> 4;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (unsigned)((size_t)a.N * a.H * a.W * C48 * 4), 0x00020000);
  // the same descriptor as raw words for the inline-asm DMA: base, stride 0, bytes, flags
  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const u32x4 rxv = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, (unsigned)((size_t)a.N * a.H * a.W * C48 * 4),
                     0x00020000u};
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (unsigned)(W_SLOTS * 16), 0x00020000);
  const unsigned out_bytes = (unsigned)((size_t)a.N * a.Ho * a.Wo * C48 * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, out_bytes, 0x00020000);

  // the filter, once (W_SLOTS is a multiple of 64: whole waves)
  for (int base = wave * 64; base < W_SLOTS; base += NTH) c48_dma16(rw, sW + base, (unsigned)(base + lane) * 16u);

  // halo slot e = it*NTH + tid -> (chunk, pixel, quad); tile-independent parts
  int rel[A_IT];  // byte offset relative to the tile's halo origin, -1 for the 12 pad pixels
  int hyx[A_IT];  // hy << 8 | hx
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int e = it * NTH + tid;
    const int c = e / CHUNK_SLOTS;
    const int rem = e - c * CHUNK_SLOTS;
    const int p = rem >> 2, q = rem & 3;
    const int hy = p / HWD, hx = p - hy * HWD;
    hyx[it] = (hy << 8) | hx;
    rel[it] = (e < HALO_SLOTS && p < HH * HWD) ? ((hy * a.W + hx) * C48 + c * EGN_CK + q * 4) * 4 : -1;
  }

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = tiles_per_img * a.N;

// Halo DMA of a tile in two parts, so that the address arithmetic of tile t+2 can be
// scheduled INTO the MFMA steps of tile t (all waves of a CU change tiles together -- one
// barrier per tile -- so every VALU instruction between the barrier and the first MFMA is
// a cycle the MFMA pipe idles): C48_OFFS computes the per-lane byte offsets (zero padding
// included: a lane outside the image gets the OOB offset, the buffer load returns 0 for it,
// so the padding is written by the same DMA), C48_ISSUE is only the A_IT DMA instructions.
#define C48_ORIGIN(T, N_, TY_, TX_)                 \
  {                                                 \
    N_ = (T) / tiles_per_img;                       \
    const int r_ = (T)-N_ * tiles_per_img;          \
    TY_ = r_ / a.tiles_x;                           \
    TX_ = r_ - TY_ * a.tiles_x;                     \
  }
#define C48_OFF1(IT, N_, TY_, TX_, OUT)                                                              \
  {                                                                                                  \
    const int iy0_ = TY_ * TH - 1, ix0_ = TX_ * TW - 1;                                              \
    const int org_ = ((N_ * a.H + iy0_) * a.W + ix0_) * C48 * 4;                                     \
    const int iy = iy0_ + (hyx[IT] >> 8), ix = ix0_ + (hyx[IT] & 255);                               \
    const bool in_ = rel[IT] >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;                     \
    OUT[IT] = in_ ? (unsigned)(org_ + rel[IT]) : EGN_OOB;                                            \
  }
#define C48_OFFS(T, OUT)                                                     \
  {                                                                          \
    int n_, ty_, tx_;                                                        \
    C48_ORIGIN(T, n_, ty_, tx_)                                              \
    _Pragma("unroll") for (int it = 0; it < A_IT; ++it) C48_OFF1(it, n_, ty_, tx_, OUT) \
  }
#define C48_ISSUE(B, OFF)                                                                            \
  {                                                                                                  \
    float4* dst_ = sA + (B)*HALO_SLOTS + wave * 64;                                                  \
    const unsigned lds_ = c48_lds_addr(dst_);                                                        \
    _Pragma("unroll") for (int it = 0; it < A_IT; ++it)                                              \
      if (it * NTH + wave * 64 < HALO_SLOTS) /* wave-uniform */                                      \
        c48_dma16_raw(rxv, lds_ + it * NTH * 16, OFF[it]);                                           \
  }
// byte offsets of this lane's MT x 4 output pixels (channel li of the first 16; + nt*64 for the
// others), EGN_OOB = masked.  EGN_OOB + 128 is still out of range: no select per access.
#define C48_VOFF1(N_, TY_, TX_, OUT)                                                                 \
  {                                                                                                  \
    const int oy0_ = TY_ * TH, ox0_ = TX_ * TW;                                                      \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                              \
      const int oy = oy0_ + wave * MT + mt;                                                          \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                \
        const int ox = ox0_ + 4 * kq + r;                                                            \
        OUT[mt][r] =                                                                                 \
            (oy < a.Ho && ox < a.Wo) ? (unsigned)(((N_ * a.Ho + oy) * a.Wo + ox) * C48 + li) * 4u : EGN_OOB; \
      }                                                                                              \
    }                                                                                                \
  }
#define C48_VOFF(T, OUT)            \
  {                                 \
    int n_, ty_, tx_;               \
    C48_ORIGIN(T, n_, ty_, tx_)     \
    C48_VOFF1(n_, ty_, tx_, OUT)    \
  }
// a piece of next-tile address arithmetic pinned between two MFMA steps: the MFMAs issued
// before it are still running in the pipe (32 cycles each) while it executes
#define C48_BETWEEN(CODE)              \
  __builtin_amdgcn_sched_barrier(0);   \
  CODE                                 \
  __builtin_amdgcn_sched_barrier(0);

  int tile = blockIdx.x;
  unsigned doff[A_IT];    // DMA offsets of the tile staged next (tile + gridDim.x inside the loop)
  unsigned voff[MT][4];   // output offsets of the current tile
  if (tile < ntiles) {
    C48_OFFS(tile, doff)
    C48_ISSUE(0, doff)
  }
  C48_OFFS(tile + (int)gridDim.x, doff)
  C48_VOFF(tile, voff)
  int buf = 0;

  // per-lane constants of the epilogue: channel = nt*16 + li, pixel x = 4*kq + r
  float sc[NT], sh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    sc[nt] = a.scale[nt * 16 + li];
    sh[nt] = a.shift[nt * 16 + li];
  }
  const int act = a.act & EGN_ACT_MASK;
  const bool res_after = (a.act & EGN_ACT_RES_AFTER) != 0;
  const bool has_res = a.res != nullptr;
  int pixbase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) pixbase[mt] = ((wave * MT + mt) * HWD + li) * EGN_CKQ + kq;

  float4 bfr[REGF ? 27 : 1][NT];
  if constexpr (REGF) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);  // the filter DMA has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s_ = 0; s_ < 27; ++s_)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bfr[s_][nt] = sW[(s_ * EGN_CKQ + kq) * C48 + nt * 16 + li];
  }

  bool first = true;
  for (; tile < ntiles; tile += gridDim.x) {
    // this tile's halo DMA (and, the first time, the filter) must have landed and the
    // the MT*NT*4 dword stores of the previous tile's epilogue are the newest vector-memory
    // operations of the lane and may stay in flight
    asm volatile("" ::: "memory");
    if (first) __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0)  expcnt(7) lgkmcnt(0)
    else if constexpr (MT == 2) __builtin_amdgcn_s_waitcnt(0x4078);  // vmcnt(24) expcnt(7) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x007C);                         // vmcnt(12) expcnt(7) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    first = false;
    const int next = tile + gridDim.x;
    if (next < ntiles) C48_ISSUE(buf ^ 1, doff)
    // the vmcnt bookkeeping above counts on program order: DMA, residual loads, stores.
    // The asm has no memory clobber (on purpose), so pin the order for the compiler here.
    asm volatile("" ::: "memory");

    // the residual values of this lane's MT x 3 x 4 outputs (without a residual every load gets
    // the OOB offset and returns 0: no branch, and the number of vector-memory operations per
    // tile -- which the vmcnt above counts on -- is fixed)
    float rv[MT][NT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned ro = has_res ? voff[mt][r] : EGN_OOB;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rv[mt][nt][r] = c48_load4(rr, ro + nt * 64u);
      }
    // address arithmetic of the tiles to come (computed in pieces between the MFMA steps below)
    unsigned doff_n[A_IT], voff_n[MT][4];
    int dn_ = 0, dty_ = 0, dtx_ = 0, vn_ = 0, vty_ = 0, vtx_ = 0;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float4* curA = sA + buf * HALO_SLOTS;
    float4 af[2][MT], bf[2][NT];
#define C48_LOADF(S, K)                                                                             \
  {                                                                                                 \
    constexpr int c_ = (S) / 9, t_ = (S) % 9;                                                       \
    constexpr int ds_ = ((t_ / 3) * HWD + (t_ % 3)) * EGN_CKQ + c_ * CHUNK_SLOTS;                   \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) af[K][mt] = curA[pixbase[mt] + ds_];          \
    if constexpr (!REGF) {                                                                          \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bf[K][nt] = sW[((S)*EGN_CKQ + kq) * C48 + nt * 16 + li]; \
    }                                                                                               \
  }
// step S with the A fragments of register set K; B from register set K or, REGF, from bfr[S]
#define C48_MFMA(K, S)                                                                               \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
    const float4 b_ = REGF ? bfr[REGF ? (S) : 0][nt] : bf[K][nt];                                    \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][mt].x, b_.x, acc[mt][nt], 0, 0, 0);     \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][mt].y, b_.y, acc[mt][nt], 0, 0, 0);     \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][mt].z, b_.z, acc[mt][nt], 0, 0, 0);     \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][mt].w, b_.w, acc[mt][nt], 0, 0, 0);     \
  }
#define C48_INTERLEAVE()                                                            \
  _Pragma("unroll") for (int k_ = 0; k_ < MT * NT * 4; ++k_) {                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x106, 1, 0);                               \
  }
// steps S and S+1: the ds_reads of the next step are in flight under the MFMAs of this one
#define C48_PAIR(S)                     \
  C48_LOADF((S) + 1, 1)                 \
  C48_MFMA(0, S)                        \
  C48_INTERLEAVE()                      \
  C48_LOADF((S) + 2 < 27 ? (S) + 2 : 26, 0) \
  C48_MFMA(1, (S) + 1)                  \
  C48_INTERLEAVE()

    static_assert(A_IT <= 9, "one C48_BETWEEN slot per DMA offset below");
    C48_LOADF(0, 0)
    C48_PAIR(0)
    C48_BETWEEN(C48_ORIGIN(next, vn_, vty_, vtx_))
    C48_PAIR(2)
    C48_BETWEEN(C48_ORIGIN(next + (int)gridDim.x, dn_, dty_, dtx_))
    C48_PAIR(4)
    C48_BETWEEN(C48_VOFF1(vn_, vty_, vtx_, voff_n))
    C48_PAIR(6)
    C48_BETWEEN(if constexpr (A_IT > 0) C48_OFF1(0, dn_, dty_, dtx_, doff_n))
    C48_PAIR(8)
    C48_BETWEEN(if constexpr (A_IT > 1) C48_OFF1(1, dn_, dty_, dtx_, doff_n))
    C48_PAIR(10)
    C48_BETWEEN(if constexpr (A_IT > 2) C48_OFF1(2, dn_, dty_, dtx_, doff_n))
    C48_PAIR(12)
    C48_BETWEEN(if constexpr (A_IT > 3) C48_OFF1(3, dn_, dty_, dtx_, doff_n))
    C48_PAIR(14)
    C48_BETWEEN(if constexpr (A_IT > 4) C48_OFF1(4, dn_, dty_, dtx_, doff_n))
    C48_PAIR(16)
    C48_BETWEEN(if constexpr (A_IT > 5) C48_OFF1(5, dn_, dty_, dtx_, doff_n))
    C48_PAIR(18)
    C48_BETWEEN(if constexpr (A_IT > 6) C48_OFF1(6, dn_, dty_, dtx_, doff_n))
    C48_PAIR(20)
    C48_BETWEEN(if constexpr (A_IT > 7) C48_OFF1(7, dn_, dty_, dtx_, doff_n))
    C48_PAIR(22)
    C48_BETWEEN(if constexpr (A_IT > 8) C48_OFF1(8, dn_, dty_, dtx_, doff_n))
    C48_PAIR(24)
    C48_MFMA(0, 26)  // step 26 (loaded by the last pair)

    // epilogue, straight from the accumulators
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[mt][nt][r] * sc[nt] + sh[nt];
          if (!res_after) v += rv[mt][nt][r];
          v = egn_act(v, act);
          if (res_after) v = rv[mt][nt][r] + v;
          c48_store4(ry, voff[mt][r] + nt * 64u, v);
        }
    buf ^= 1;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) doff[it] = doff_n[it];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) voff[mt][r] = voff_n[mt][r];
  }
#undef C48_ORIGIN
#undef C48_OFF1
#undef C48_OFFS
#undef C48_ISSUE
#undef C48_VOFF1
#undef C48_VOFF
#undef C48_BETWEEN
#undef C48_LOADF
#undef C48_MFMA
#undef C48_INTERLEAVE
#undef C48_PAIR
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void conv_c48_kernel(ConvArgs a) {
  c48_body<WAVES, false>(a);
}
// the register-resident-filter variant (config 43)
#ifdef EGN_PROBES   // cfg 43 (register-resident filter): measured, never selected -- probe builds only
__global__ __launch_bounds__(256, 1) void conv_c48r_kernel(ConvArgs a) { c48_body<4, true>(a); }
#endif


// ---------------------------------------------------------------------------------------------
// conv_c48t_kernel -- the same layers on a 16 x 16 pixel tile (config 44): 8 waves x TWO tile rows,
// so a K step is 24 MFMAs per 5 ds_read_b128 instead of 12 per 4 (the inner loop in isolation,
// tools/micro/mfma_lds.hip: 133 vs 120 TFLOP/s).  A double-buffered 18 x 18 x 48 halo does not
// fit next to the filter, so the halo is kept as a RING OF ITS THREE 16-CHANNEL CHUNKS
// (3 x 21.5 KB + 82.9 KB = 147.5 KB): chunk c is only read during K steps 9c .. 9c+8, and as soon
// as every wave is past them (one barrier per chunk) the next tile's chunk c is DMA'd into the
// same buffer -- two thirds of a tile ahead of its first use.
//   top of tile t : wait D(t,c0), D(t,c1); barrier; issue D(t,c2); residual loads; steps 0..7
//   barrier 1     : issue D(t+1,c0);                                steps 8..16
//   wait D(t,c2); barrier 2 : issue D(t+1,c1);                      steps 17..26; epilogue stores
// (a step's fragments are fetched during the step before it, so the barrier that frees chunk c
//  sits in front of the last step of chunk c, after its fragment reads have completed)
// vmcnt bookkeeping (program order, in-order completion): at the top the newest 24 operations are
// the stores of tile t-1, everything older (both chunk DMAs) must have landed -> vmcnt(24); before
// barrier 2 the newest >= 24 are the residual loads (+ the DMA just issued) -> vmcnt(24) again.
namespace {
constexpr int T_TH = 16, T_HH = T_TH + 2;
constexpr int T_NPIX = T_HH * HWD;                 // 324 halo pixels
constexpr int T_NPIXP = 336;                       // rounded up: CHUNK slots are whole waves
constexpr int T_CHUNK_SLOTS = T_NPIXP * EGN_CKQ;   // 1344 float4 = 21.5 KB
constexpr int T_WAVES = 8, T_NTH = 64 * T_WAVES, T_MT = 2;
constexpr int T_IT = (T_CHUNK_SLOTS + T_NTH - 1) / T_NTH;  // 3 DMA instructions per lane and chunk
}  // namespace

__global__ __launch_bounds__(T_NTH, 1) void conv_c48t_kernel(ConvArgs a) {
  extern __shared__ float4 smem[];
  float4* sW = smem;             // [chunk][tap][quad][48]
  float4* sA = smem + W_SLOTS;   // [chunk][pixel][quad]: one buffer per chunk

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;
  const int kq = lane >> 4;

  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.x);
  const u32x4 rxv = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, (unsigned)((size_t)a.N * a.H * a.W * C48 * 4),
                     0x00020000u};
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (unsigned)(W_SLOTS * 16), 0x00020000);
  const unsigned out_bytes = (unsigned)((size_t)a.N * a.Ho * a.Wo * C48 * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.y), 0, out_bytes, 0x00020000);

  for (int base = wave * 64; base < W_SLOTS; base += T_NTH) c48_dma16(rw, sW + base, (unsigned)(base + lane) * 16u);

  // chunk slot e = it*NTH + tid -> (pixel, quad); chunk c adds c*64 bytes to the source offset
  int rel[T_IT], hyx[T_IT];
#pragma unroll
  for (int it = 0; it < T_IT; ++it) {
    const int e = it * T_NTH + tid;
    const int p = e >> 2, q = e & 3;
    const int hy = p / HWD, hx = p - hy * HWD;
    hyx[it] = (hy << 8) | hx;
    rel[it] = (e < T_CHUNK_SLOTS && p < T_NPIX) ? ((hy * a.W + hx) * C48 + q * 4) * 4 : -1;
  }
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = tiles_per_img * a.N;

#define T_ORIGIN(T, N_, TY_, TX_)                   \
  {                                                 \
    N_ = (T) / tiles_per_img;                       \
    const int r_ = (T)-N_ * tiles_per_img;          \
    TY_ = r_ / a.tiles_x;                           \
    TX_ = r_ - TY_ * a.tiles_x;                     \
  }
#define T_OFF1(IT, N_, TY_, TX_, OUT)                                                                \
  {                                                                                                  \
    const int iy0_ = TY_ * T_TH - 1, ix0_ = TX_ * TW - 1;                                            \
    const int org_ = ((N_ * a.H + iy0_) * a.W + ix0_) * C48 * 4;                                     \
    const int iy = iy0_ + (hyx[IT] >> 8), ix = ix0_ + (hyx[IT] & 255);                               \
    const bool in_ = rel[IT] >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;                     \
    OUT[IT] = in_ ? (unsigned)(org_ + rel[IT]) : EGN_OOB;                                            \
  }
// chunk C of the tile whose chunk-0 offsets are OFF, into ring buffer C (EGN_OOB + 128 stays OOB)
#define T_ISSUE(C, OFF)                                                                              \
  {                                                                                                  \
    const unsigned lds_ = c48_lds_addr(sA + (C)*T_CHUNK_SLOTS + wave * 64);                          \
    _Pragma("unroll") for (int it = 0; it < T_IT; ++it)                                              \
      if (it * T_NTH + wave * 64 < T_CHUNK_SLOTS) /* wave-uniform */                                 \
        c48_dma16_raw(rxv, lds_ + it * T_NTH * 16, OFF[it] + (C)*64u);                               \
  }
#define T_VOFF1(N_, TY_, TX_, OUT)                                                                   \
  {                                                                                                  \
    const int oy0_ = TY_ * T_TH, ox0_ = TX_ * TW;                                                    \
    _Pragma("unroll") for (int mt = 0; mt < T_MT; ++mt) {                                            \
      const int oy = oy0_ + wave * T_MT + mt;                                                        \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                \
        const int ox = ox0_ + 4 * kq + r;                                                            \
        OUT[mt][r] =                                                                                 \
            (oy < a.Ho && ox < a.Wo) ? (unsigned)(((N_ * a.Ho + oy) * a.Wo + ox) * C48 + li) * 4u : EGN_OOB; \
      }                                                                                              \
    }                                                                                                \
  }
#define T_BETWEEN(CODE)                \
  __builtin_amdgcn_sched_barrier(0);   \
  CODE                                 \
  __builtin_amdgcn_sched_barrier(0);

  int tile = blockIdx.x;
  unsigned doff[T_IT];     // chunk-0 DMA offsets of the CURRENT tile (its chunk 2 is issued at the top)
  unsigned doff_n[T_IT];   // ... of the next tile
  unsigned voff[T_MT][4];
  {
    int n_ = 0, ty_ = 0, tx_ = 0;
    T_ORIGIN(tile, n_, ty_, tx_)
#pragma unroll
    for (int it = 0; it < T_IT; ++it) T_OFF1(it, n_, ty_, tx_, doff)
    T_VOFF1(n_, ty_, tx_, voff)
    T_ORIGIN(tile + (int)gridDim.x, n_, ty_, tx_)
#pragma unroll
    for (int it = 0; it < T_IT; ++it) T_OFF1(it, n_, ty_, tx_, doff_n)
  }
  if (tile < ntiles) {
    T_ISSUE(0, doff)
    T_ISSUE(1, doff)
  }

  float sc[NT], sh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    sc[nt] = a.scale[nt * 16 + li];
    sh[nt] = a.shift[nt * 16 + li];
  }
  const int act = a.act & EGN_ACT_MASK;
  const bool res_after = (a.act & EGN_ACT_RES_AFTER) != 0;
  const bool has_res = a.res != nullptr;
  int pixbase[T_MT];
#pragma unroll
  for (int mt = 0; mt < T_MT; ++mt) pixbase[mt] = ((wave * T_MT + mt) * HWD + li) * EGN_CKQ + kq;

  bool first = true;
  for (; tile < ntiles; tile += gridDim.x) {
    asm volatile("" ::: "memory");
    if (first) __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): filter, chunks 0 and 1
    else __builtin_amdgcn_s_waitcnt(0x4078);        // vmcnt(24): all but the previous tile's stores
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    first = false;
    T_ISSUE(2, doff)
    asm volatile("" ::: "memory");
    const int next = tile + gridDim.x;

    float rv[T_MT][NT][4];
#pragma unroll
    for (int mt = 0; mt < T_MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned ro = has_res ? voff[mt][r] : EGN_OOB;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rv[mt][nt][r] = c48_load4(rr, ro + nt * 64u);
      }
    unsigned doff_nn[T_IT], voff_n[T_MT][4];
    int dn_ = 0, dty_ = 0, dtx_ = 0, vn_ = 0, vty_ = 0, vtx_ = 0;

    f32x4 acc[T_MT][NT];
#pragma unroll
    for (int mt = 0; mt < T_MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 af[2][T_MT], bf[2][NT];
#define T_LOADF(S, K)                                                                               \
  {                                                                                                 \
    constexpr int c_ = (S) / 9, t_ = (S) % 9;                                                       \
    constexpr int ds_ = ((t_ / 3) * HWD + (t_ % 3)) * EGN_CKQ + c_ * T_CHUNK_SLOTS;                 \
    _Pragma("unroll") for (int mt = 0; mt < T_MT; ++mt) af[K][mt] = sA[pixbase[mt] + ds_];          \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bf[K][nt] = sW[((S)*EGN_CKQ + kq) * C48 + nt * 16 + li]; \
  }
#define T_MFMA(K)                                                                                    \
  _Pragma("unroll") for (int mt = 0; mt < T_MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][mt].x, bf[K][nt].x, acc[mt][nt], 0, 0, 0); \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][mt].y, bf[K][nt].y, acc[mt][nt], 0, 0, 0); \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][mt].z, bf[K][nt].z, acc[mt][nt], 0, 0, 0); \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][mt].w, bf[K][nt].w, acc[mt][nt], 0, 0, 0); \
  }
// step S: the fragments of step S+1 are fetched under its MFMAs
#if defined(C48T_INTERLEAVE)
#define T_INTER()                                                      \
  _Pragma("unroll") for (int k_ = 0; k_ < T_MT * NT * 4; ++k_) {       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
    __builtin_amdgcn_sched_group_barrier(0x106, 1, 0);                 \
  }
#elif defined(C48T_LOADS_FIRST)
#define T_INTER()                                                      \
  __builtin_amdgcn_sched_group_barrier(0x100, T_MT + NT, 0);           \
  __builtin_amdgcn_sched_group_barrier(0x008, T_MT * NT * 4, 0);
#else
#define T_INTER()
#endif
#define T_STEP(S)                                             \
  T_LOADF((S) + 1 < 27 ? (S) + 1 : 26, ((S) + 1) & 1)         \
  T_MFMA((S)&1)                                               \
  T_INTER()

    T_LOADF(0, 0)
    T_STEP(0) T_STEP(1)
    T_BETWEEN(T_ORIGIN(next, vn_, vty_, vtx_))
    T_STEP(2) T_STEP(3)
    T_BETWEEN(T_VOFF1(vn_, vty_, vtx_, voff_n))
    T_STEP(4) T_STEP(5) T_STEP(6) T_STEP(7)
    // step 8's fragments (the last read of chunk 0) are in flight: once they are in registers
    // -- lgkmcnt(0) -- and every wave is here, the next tile's chunk 0 may overwrite the buffer
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // vmcnt(63) expcnt(7) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (next < ntiles) T_ISSUE(0, doff_n)
    asm volatile("" ::: "memory");
    T_STEP(8) T_STEP(9) T_STEP(10)
    T_BETWEEN(T_ORIGIN(next + (int)gridDim.x, dn_, dty_, dtx_))
    T_STEP(11) T_STEP(12)
    T_BETWEEN(T_OFF1(0, dn_, dty_, dtx_, doff_nn))
    T_STEP(13) T_STEP(14)
    T_BETWEEN(T_OFF1(1, dn_, dty_, dtx_, doff_nn))
    T_STEP(15) T_STEP(16)
    T_BETWEEN(T_OFF1(2, dn_, dty_, dtx_, doff_nn))
    // Before step 17 (whose fragments are already in registers) fetches step 18 from chunk 2:
    // this tile's chunk 2 (issued at the top) must have landed -- vmcnt(24), every wave's share,
    // hence the barrier -- and every wave's last read of chunk 1 (step 17's fragments) must be
    // complete -- lgkmcnt(0) -- before the next tile's chunk 1 overwrites it.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x4078);  // vmcnt(24) expcnt(7) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (next < ntiles) T_ISSUE(1, doff_n)
    asm volatile("" ::: "memory");
    T_STEP(17)
    T_STEP(18) T_STEP(19) T_STEP(20) T_STEP(21) T_STEP(22) T_STEP(23) T_STEP(24) T_STEP(25)
    T_MFMA(26 & 1)

#pragma unroll
    for (int mt = 0; mt < T_MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[mt][nt][r] * sc[nt] + sh[nt];
          if (!res_after) v += rv[mt][nt][r];
          v = egn_act(v, act);
          if (res_after) v = rv[mt][nt][r] + v;
          c48_store4(ry, voff[mt][r] + nt * 64u, v);
        }
#pragma unroll
    for (int it = 0; it < T_IT; ++it) {
      doff[it] = doff_n[it];
      doff_n[it] = doff_nn[it];
    }
#pragma unroll
    for (int mt = 0; mt < T_MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) voff[mt][r] = voff_n[mt][r];
  }
#undef T_ORIGIN
#undef T_OFF1
#undef T_ISSUE
#undef T_VOFF1
#undef T_BETWEEN
#undef T_LOADF
#undef T_MFMA
#undef T_STEP
#undef T_INTER
}

template <int WAVES>
static int c48_launch(const ConvArgs& a, size_t lds, int grid, hipStream_t stream) {
  static bool raised[EGN_MAX_DEVICES];  // per instantiation and device
  if (egn_first_use_on_device(raised)) {
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c48_kernel<WAVES>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  }
  hipLaunchKernelGGL((conv_c48_kernel<WAVES>), dim3(grid), dim3(64 * WAVES), lds, stream, a);
  return (int)hipGetLastError();
}

int egn_conv_launch_c48(const ConvArgs& a, size_t lds, int waves, hipStream_t stream) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int ntiles = a.tiles_x * a.tiles_y * a.N;
  const int grid = ntiles < cus ? ntiles : cus;
  if (waves == -1) {  // 16 x 16 tile, halo as a ring of chunks
    static bool raised_t[EGN_MAX_DEVICES];
    if (egn_first_use_on_device(raised_t)) {
      EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c48t_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    }
    hipLaunchKernelGGL(conv_c48t_kernel, dim3(grid), dim3(512), lds, stream, a);
    return (int)hipGetLastError();
  }
#ifdef EGN_PROBES
  if (waves == 0) {  // register-resident filter
    static bool raised[EGN_MAX_DEVICES];
    if (egn_first_use_on_device(raised)) {
      EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c48r_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    }
    hipLaunchKernelGGL(conv_c48r_kernel, dim3(grid), dim3(256), lds, stream, a);
    return (int)hipGetLastError();
  }
  if (waves == 4) return c48_launch<4>(a, lds, grid, stream);       // cfg 41: never selected either
#endif
  return waves == 8 ? c48_launch<8>(a, lds, grid, stream) : EGN_E_BADARG;
}
