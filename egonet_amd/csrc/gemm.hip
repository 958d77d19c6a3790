// gemm.hip -- fp32 GEMM on the matrix pipe for the lifter's dense layers (reference
// libs/model/FCmodel.py:33-43, 92-105: six nn.Linear per forward; libs/trainer/trainer.py:191-197 adds their
// data and weight gradients), written from scratch for gfx950.
//
// Until round 3 these ran as 1x1 convolutions on the conv kernels (K chunks of 16, one barrier per 32 MFMAs,
// a prepacked weight blob): 88 TFLOP/s on 4096 x 1024 x 1024.  A dense GEMM has no halo and no taps, so the K
// stage can be deep and the operands can be read from HBM exactly as they lie:
//
//   C[M][N] = sum_k opA(m,k) * opB(k,n)  (+ bias[n])
//
//   operand layouts (row-major, leading dimension in floats):
//     "KC"  K-contiguous:  A[m][k] (lda)  /  B[n][k] (ldb)      -- activations x W^T (forward: z = a W^T)
//     "MC"  M-contiguous:  A[k][m] (lda)  /  B[k][n] (ldb)      -- dz^T a (weight gradient), dz W (data gradient)
//   instantiated: NT (A KC, B KC) forward; NN (A KC, B MC) data gradient with W as it lies; TN (A MC, B MC)
//   weight gradient with both activations as they lie, split along K (= the batch) with a fixed-order
//   reduction of the partial slabs -- nothing is transposed or packed in HBM.
//
//   * block tile BM x BN x 32, v_mfma_f32_16x16x4_f32, wave tile (BM/WM) x (BN/WN);
//   * K stages arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`, raw ISA like csrc/conv_dma.hip) into a ring of
//     STAGES buffers, ONE barrier per 32-deep stage, counted `s_waitcnt vmcnt(n)`: the stage after next stays in
//     flight across the barrier;
//   * the LDS image of a DMA is lane-linear, so the layout is made by the SOURCE address:
//       KC tile  [row][8 quads of 4 k], quad' = quad ^ ((row >> 1) & 7): every 16-lane group of a fragment
//                ds_read_b128 hits 16 distinct 16-byte columns (checked for all four lane groups in
//                tests/test_gemm_design_cpu.py), while each DMA instruction still reads 8 rows x 128 contiguous bytes;
//       MC tile  [k][rows], rows contiguous as in HBM (512 contiguous bytes per k); a lane's b128 holds FOUR
//                rows 4i..4i+3 of one k, which become four interleaved 16-row MFMA tiles (row = 4*idx + j);
//   * fragment mapping: lane (i = l & 15, kq = l >> 4); per 16 k:
//       KC x KC: a[s] = A[i][4kq+s], b[s] = B[n][4kq+s], MFMA s                      -> C[16mt + 4kq + r][16nt + i]
//       KC x MC: a[s] as above, b_s[j] = B[4kq+s][4i+j], MFMA (s, j)                 -> C[16mt + 4kq + r][4i + j]
//       MC x MC: a[ja] = A[k0+kq][4i+ja], b[jb] = B[k0+kq][4i+jb], MFMA (ja, jb)     -> C[4(4kq+r) + ja][4i + jb]
//     so with an MC B operand a lane owns 4 consecutive columns: 16-byte stores straight from the accumulators;
//     the NT form goes through a wave-private LDS transpose for the same store width;
//   * block index -> tile: XCD-aware (block b runs on XCD b % 8): the blocks of one XCD share rows of A.
//
// Shapes: M % BM == 0, N % BN == 0, K % 32 == 0 (per split), leading dimensions % 4 == 0; anything else stays on
// the conv-kernel route (egn_gemm_supported).  Numerics: an fp32 fmaf chain in k order per split, splits added
// in a fixed order -- deterministic.
#include <stdlib.h>

#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_gemm_t;

namespace {

constexpr int GK = 32;  // k per stage

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;           // [M][ldc], or the partial slabs [splits][M][N] when splits > 1
  const float* bias;  // [N] or null
  // [r4] epilogue fusions of the lifter step (libs/model/FCmodel.py:33-43 under trainer.py:183-209):
  const float* addend;  // NN form: C = A.B + addend (same ldc): the skip-path gradient of a residual block, no add pass
  double* stats;        // NT form: partial column sums / sums of squares of the stored C, [tiles_m][2][N] doubles --
                        // BatchNorm1d's batch statistics without a pass over z (finalised by egn_bn_stats_finalize_f32)
  int M, N, K;        // K = k range of ONE split
  int lda, ldb, ldc;
  int tiles_m, tiles_n, splits;
  int raster;         // block -> (split, tile) order, see the kernel (EGONET_AMD_GEMM_RASTER=0: round 4's order)
  size_t a_bytes, b_bytes;
};

__device__ __forceinline__ void gemm_dma16(u32x4 r, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(r), "s"(soff)
               : "m0");
}

template <int N>
__device__ __forceinline__ void gemm_wait_vm() {
  static_assert(N >= 0 && N <= 16, "vmcnt");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "add the literal");
}

// byte offset of (row, quad) in a KC tile image (128 B per row, XOR-swizzled quads)
__device__ __forceinline__ unsigned kc_off(int row, int quad) { return (unsigned)row * 128u + (unsigned)((quad ^ ((row >> 1) & 7)) << 4); }

template <bool AKC, bool BKC, int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void gemm_kernel(GemmArgs g) {
  constexpr int NW = WM * WN;
  constexpr int NTH = 64 * NW;
  constexpr int TM = BM / WM, TN = BN / WN;    // wave tile
  constexpr int MT = TM / 16, NT = TN / 16;
  static_assert(AKC || TM % 64 == 0, "an M-contiguous A operand feeds four interleaved 16-row tiles");
  static_assert(BKC || TN % 64 == 0, "an N-contiguous B operand feeds four interleaved 16-column tiles");
  static_assert(AKC || !BKC, "TN, NN and NT are built; the fourth combination has no caller");
  constexpr int A_BYTES = BM * GK * 4, B_BYTES = BN * GK * 4, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024, P = PA + PB;   // DMA instructions per stage
  static_assert(P % NW == 0, "whole DMA pieces per wave");
  constexpr int PW = P / NW;
  extern __shared__ float4 smem[];
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_gemm_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, kq = lane >> 4;

  // block -> (split, tile).  b % 8 = XCD: the blocks of an XCD take consecutive n-tiles of the same m-tile rows
  // (their A rows are fetched into that XCD's L2 once; B is small enough to be resident everywhere)
  const int ntiles = g.tiles_m * g.tiles_n;
  const int per = (ntiles + 7) >> 3;                 // tile slots per XCD and split (the launcher's grid = 8 * per * splits)
  const int b = blockIdx.x;
  int split, t;
  if (g.raster) {
    // [round 5] the XCD owns a contiguous run of the list of (split, tile) pairs: with split K (the weight gradient:
    // 4 splits x 64 tiles) an XCD works on ONE K slice and four rows of tiles -- 2 MB of A and 4 MB of B per XCD
    // instead of a panel of A and all of B of EVERY slice (profiles/r4_pmc_traffic.json: 168 MB per launch against
    // 37.7 MB algorithmic).  With one split this is the order of the other branch.
    const int x = b & 7, q = b >> 3;                 // XCD, slot on it (per * splits slots)
    const int l = x * per * g.splits + q;
    split = l / ntiles;
    t = l - split * ntiles;
    if (split >= g.splits) return;                   // (ntiles % 8 != 0: rounded-up grid)
  } else {
    split = b / (8 * per);
    t = b - split * 8 * per;
    const int x = t & 7, q = t >> 3;                 // XCD, slot on it
    t = x * per + q;                                 // tiles [x*per, (x+1)*per) live on XCD x
    if (t >= ntiles) return;                         // (ntiles % 8 != 0: rounded-up grid)
  }
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * g.K;

  const unsigned long long aaddr = reinterpret_cast<unsigned long long>(g.A), baddr = reinterpret_cast<unsigned long long>(g.B);
  const u32x4 ra = {(unsigned)aaddr, (unsigned)(aaddr >> 32) & 0xffffu, (unsigned)g.a_bytes, 0x00020000u};
  const u32x4 rb = {(unsigned)baddr, (unsigned)(baddr >> 32) & 0xffffu, (unsigned)g.b_bytes, 0x00020000u};

  // this wave's DMA pieces of a stage: piece p = j * NW + wave; p < PA: A tile, else B tile
  unsigned pvoff[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int p = j * NW + wave;
    const bool isA = p < PA;
    const int q = isA ? p : p - PA;
    const bool kc = isA ? AKC : BKC;
    const int ld = isA ? g.lda : g.ldb;
    const int r0 = isA ? m0 : n0;
    if (kc) {            // 8 rows x 8 quads per piece
      const int row = q * 8 + (lane >> 3), quad = (lane & 7) ^ ((row >> 1) & 7);
      pvoff[j] = (unsigned)((size_t)(r0 + row) * ld * 4 + (size_t)(kbeg + quad * 4) * 4);
    } else {             // k-rows of (BM or BN) floats; 64 float4 per piece
      constexpr int dummy = 0;
      const int w4 = (isA ? BM : BN) / 4;                // float4 per k-row
      const int e = q * 64 + lane;
      const int krow = e / w4, c4 = e - krow * w4;
      pvoff[j] = (unsigned)((size_t)(kbeg + krow) * ld * 4 + (size_t)(r0 + c4 * 4) * 4);
      (void)dummy;
    }
  }
  const unsigned a_kstep = AKC ? (unsigned)GK * 4u : (unsigned)GK * (unsigned)g.lda * 4u;   // bytes per stage along k
  const unsigned b_kstep = BKC ? (unsigned)GK * 4u : (unsigned)GK * (unsigned)g.ldb * 4u;

#define GEMM_ISSUE(STG, KS)                                                                                  \
  {                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < PW; ++j) {                                                         \
      const int p = j * NW + wave;                                                                           \
      const bool isA = p < PA;                                                                               \
      gemm_dma16(isA ? ra : rb, lds0 + (unsigned)(STG)*ST_BYTES + (unsigned)p * 1024u, pvoff[j],             \
                 (unsigned)(KS) * (isA ? a_kstep : b_kstep));                                                \
    }                                                                                                        \
  }

  const int nks = g.K / GK;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nks) GEMM_ISSUE(s, s)

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment base offsets inside a stage image
  const unsigned a_row = (unsigned)(wm * TM), b_row = (unsigned)(wn * TN);
  int stg = 0;
  for (int ks = 0; ks < nks; ++ks) {
    // stage ks landed (this wave's pieces) -- the next stage's pieces may stay in flight -- then everyone's
    asm volatile("" ::: "memory");
    if (ks + STAGES - 2 < nks) gemm_wait_vm<(STAGES - 2) * PW>(); else gemm_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // refill the buffer the previous step consumed (every wave is past its reads: the barrier above)
    {
      const int nk = ks + STAGES - 1;
      int nstg = stg + STAGES - 1;
      if (nstg >= STAGES) nstg -= STAGES;
      if (nk < nks) GEMM_ISSUE(nstg, nk)
    }
    const char* sa = reinterpret_cast<const char*>(smem) + stg * ST_BYTES;
    const char* sb = sa + A_BYTES;
    if constexpr (AKC && BKC) {
#pragma unroll
      for (int grp = 0; grp < 2; ++grp) {
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(sa + kc_off(a_row + i * 16 + li, 4 * grp + kq));
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(sb + kc_off(b_row + j * 16 + li, 4 * grp + kq));
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (AKC && !BKC) {
#pragma unroll
      for (int grp = 0; grp < 2; ++grp) {
        f32x4 af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(sa + kc_off(a_row + i * 16 + li, 4 * grp + kq));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          f32x4 bq[NT / 4];
#pragma unroll
          for (int q = 0; q < NT / 4; ++q)
            bq[q] = *reinterpret_cast<const f32x4*>(sb + (unsigned)(16 * grp + 4 * kq + s) * (BN * 4) + (b_row + q * 64 + 4 * li) * 4);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NT / 4; ++q)
#pragma unroll
              for (int jb = 0; jb < 4; ++jb)
                acc[i][q * 4 + jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bq[q][jb], acc[i][q * 4 + jb], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < GK / 4; ++kk) {
        f32x4 aq[MT / 4], bq[NT / 4];
#pragma unroll
        for (int q = 0; q < MT / 4; ++q)
          aq[q] = *reinterpret_cast<const f32x4*>(sa + (unsigned)(4 * kk + kq) * (BM * 4) + (a_row + q * 64 + 4 * li) * 4);
#pragma unroll
        for (int q = 0; q < NT / 4; ++q)
          bq[q] = *reinterpret_cast<const f32x4*>(sb + (unsigned)(4 * kk + kq) * (BN * 4) + (b_row + q * 64 + 4 * li) * 4);
#pragma unroll
        for (int qa = 0; qa < MT / 4; ++qa)
#pragma unroll
          for (int ja = 0; ja < 4; ++ja)
#pragma unroll
            for (int qb = 0; qb < NT / 4; ++qb)
#pragma unroll
              for (int jb = 0; jb < 4; ++jb)
                acc[qa * 4 + ja][qb * 4 + jb] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(aq[qa][ja], bq[qb][jb], acc[qa * 4 + ja][qb * 4 + jb], 0, 0, 0);
      }
    }
    if (++stg == STAGES) stg = 0;
  }
#undef GEMM_ISSUE

  // ---- epilogue: 16-byte stores along n ----
  float* Cb = g.C + (g.splits > 1 ? (size_t)split * g.M * g.ldc : 0);
  if constexpr (!BKC) {
    // lane owns 4 consecutive columns 4li + jb of every 64-column group: straight from the accumulators
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int q = 0; q < NT / 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row;
          if constexpr (AKC) row = m0 + wm * TM + i * 16 + 4 * kq + r;
          else row = m0 + wm * TM + (i >> 2) * 64 + 4 * (4 * kq + r) + (i & 3);
          const int col = n0 + wn * TN + q * 64 + 4 * li;
          f32x4 v = {acc[i][q * 4 + 0][r], acc[i][q * 4 + 1][r], acc[i][q * 4 + 2][r], acc[i][q * 4 + 3][r]};
          if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
          if (g.addend) v += *reinterpret_cast<const f32x4*>(g.addend + (size_t)row * g.ldc + col);
          *reinterpret_cast<f32x4*>(Cb + (size_t)row * g.ldc + col) = v;
        }
  } else {
    // NT: lane owns column li of every 16-column tile -> transpose through a wave-private LDS slab
    constexpr int SLD = TN + 4;                       // floats per slab row (16-byte aligned, spreads banks)
    static_assert(NW * 16 * SLD * 4 <= STAGES * ST_BYTES, "the transpose slabs fit in the stage ring");
    __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): this wave's fragment reads are done
    __builtin_amdgcn_s_barrier();                     // ... and everyone else's (the slabs alias the stage ring)
    float* sC = reinterpret_cast<float*>(smem) + wave * 16 * SLD;
    constexpr int C4 = TN / 4;                        // float4 per slab row
    // column statistics of what is stored (z = a W^T + b): lane (li, kq) owns column 16 j + li of its wave tile for the
    // rows 16 i + 4 kq + r -- sums in fp32 per lane (4 MT values), across kq by shuffles, across the WM waves of a
    // column in doubles through a table behind the slabs, in a fixed order; the block writes ITS columns of row tm
    double* sS = reinterpret_cast<double*>(reinterpret_cast<char*>(smem) + NW * 16 * SLD * 4);
    static_assert(NW * 16 * SLD * 4 + NW * 2 * TN * 8 <= STAGES * ST_BYTES, "slabs + statistics table fit in the stage ring");
    if (g.stats != nullptr) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float bj = g.bias ? g.bias[n0 + wn * TN + j * 16 + li] : 0.f;
        // [round 5, ADVICE r4] in doubles from the first add: with fp32 squares a column whose |mean| is 1e3 x its
        // deviation lost several digits of E[x^2] - mean^2 (tests/test_gpu_gemm.py: the large-mean case)
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double v = (double)(acc[i][j][r] + bj);
            s1 += v;
            s2 += v * v;
          }
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        if (kq == 0) {
          sS[(wave * 2 + 0) * TN + j * 16 + li] = s1;
          sS[(wave * 2 + 1) * TN + j * 16 + li] = s2;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[(4 * kq + r) * SLD + j * 16 + li] = acc[i][j][r];
      __builtin_amdgcn_s_waitcnt(0xC07F);             // wave-private: own writes visible to own reads after lgkmcnt(0)
#pragma unroll
      for (int e = lane; e < 16 * C4; e += 64) {
        const int rr = e / C4, c4 = e - rr * C4;
        f32x4 v = *reinterpret_cast<const f32x4*>(&sC[rr * SLD + c4 * 4]);
        const int row = m0 + wm * TM + i * 16 + rr, col = n0 + wn * TN + c4 * 4;
        if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
        *reinterpret_cast<f32x4*>(Cb + (size_t)row * g.ldc + col) = v;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);             // reads done before the next tile row overwrites the slab
    }
    if (g.stats != nullptr) {
      __syncthreads();
      for (int e = tid; e < 2 * BN; e += NTH) {
        const int which = e / BN, c = e - which * BN;
        const int cwn = c / TN, cl = c - cwn * TN;
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < WM; ++k) v += sS[((k * WN + cwn) * 2 + which) * TN + cl];
        g.stats[((size_t)tm * 2 + which) * g.N + n0 + c] = v;
      }
    }
  }
}

// C[m][n] = sum over the split slabs in a fixed order (+ nothing): 16 B per lane
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ C,
                                                                 int M, int N, int ldc, int splits) {
  const size_t total4 = (size_t)M * N / 4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
    const size_t row = e / (N / 4), c4 = e - row * (N / 4);
    f32x4 v = *reinterpret_cast<const f32x4*>(part + e * 4);
    for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(part + ((size_t)s * M * N) + e * 4);
    *reinterpret_cast<f32x4*>(C + row * ldc + c4 * 4) = v;
  }
}

template <bool AKC, bool BKC, int BM, int BN, int WM, int WN, int STAGES>
int gemm_launch(GemmArgs g, hipStream_t st) {
  static bool raised[EGN_MAX_DEVICES];
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * GK * 4;
  auto k = &gemm_kernel<AKC, BKC, BM, BN, WM, WN, STAGES>;
  if (egn_first_use_on_device(raised))
    EGN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g.tiles_m = g.M / BM;
  g.tiles_n = g.N / BN;
  static const int raster = [] { const char* e = getenv("EGONET_AMD_GEMM_RASTER"); return e && e[0] == '0' ? 0 : 1; }();
  g.raster = raster;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int grid = ((ntiles + 7) / 8) * 8 * g.splits;      // 8 * per tile slots per split (kernel: XCD remap)
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WM * WN), lds, st, g);
  return (int)hipGetLastError();
}

}  // namespace

// variant: tile configuration (0 = default of the form); see egn_gemm_variants
//   form 0 NT: C = A[M][K] B[N][K]^T (+bias)     form 1 NN: C = A[M][K] B[K][N]     form 2 TN: C = A[K][M]^T B[K][N]
extern "C" int egn_gemm_supported(int form, int M, int N, int K, int lda, int ldb, int ldc) {
  if (form < 0 || form > 2 || M <= 0 || N <= 0 || K <= 0) return 0;
  if ((lda | ldb | ldc) & 3) return 0;
  if (M % 128 || N % 128 || K % GK) return 0;
  if (form == 2 && ((M / 128) * (N / 128)) % 8) return 0;
  return 1;
}

extern "C" long egn_gemm_ws_bytes(int form, int M, int N, int K) {
  if (form != 2) return 0;
  const int tiles = (M / 128) * (N / 128);
  int splits = 1;
  while (tiles * splits < 256 && (K / (splits * 2)) % GK == 0 && K / (splits * 2) >= 4 * GK) splits *= 2;
  return splits > 1 ? (long)splits * M * N * 4 : 0;
}

// rows of the partial statistics table of egn_gemm_ex_f32(form 0, stats): one per 128-row block tile
extern "C" long egn_gemm_stats_rows(int M) { return M > 0 && M % 128 == 0 ? M / 128 : 0; }

extern "C" int egn_gemm_f32(int form, const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                            int lda, int ldb, int ldc, int variant, void* ws, long ws_bytes, void* stream) {
  return egn_gemm_ex_f32(form, A, B, C, bias, nullptr, nullptr, 0, M, N, K, lda, ldb, ldc, variant, ws, ws_bytes, stream);
}

extern "C" int egn_gemm_ex_f32(int form, const float* A, const float* B, float* C, const float* bias, const float* addend,
                               double* stats, long stats_rows, int M, int N, int K, int lda, int ldb, int ldc, int variant,
                               void* ws, long ws_bytes, void* stream) {
  if (!A || !B || !C || !egn_gemm_supported(form, M, N, K, lda, ldb, ldc)) return EGN_E_BADARG;
  if (addend && form != 1) return EGN_E_BADARG;                                  // C = A.B + addend: the NN form only
  if (stats && (form != 0 || stats_rows < egn_gemm_stats_rows(M))) return EGN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g = {};
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.addend = addend; g.stats = stats;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.splits = 1;
  g.a_bytes = (size_t)(form == 2 ? K : M) * lda * 4;
  g.b_bytes = (size_t)(form == 0 ? N : K) * ldb * 4;
  if (g.a_bytes >= 0xF0000000ull || g.b_bytes >= 0xF0000000ull) return EGN_E_BADARG;
  if (form == 0) {
    switch (variant) {
      case 1: return gemm_launch<true, true, 128, 128, 2, 2, 3>(g, st);
      case 2: return gemm_launch<true, true, 128, 64, 2, 2, 3>(g, st);
      case 3: return gemm_launch<true, true, 128, 128, 2, 4, 2>(g, st);
      default: return gemm_launch<true, true, 128, 128, 2, 4, 3>(g, st);
    }
  }
  if (form == 1) {
    switch (variant) {
      case 1: return gemm_launch<true, false, 128, 128, 2, 2, 3>(g, st);
      case 2: return gemm_launch<true, false, 128, 64, 2, 1, 3>(g, st);
      default: return gemm_launch<true, false, 128, 128, 4, 2, 3>(g, st);
    }
  }
  // TN: split K until the grid fills the chip; partial slabs + fixed-order reduction
  const long need = egn_gemm_ws_bytes(form, M, N, K);
  if (bias) return EGN_E_BADARG;
  int splits = need ? (int)(need / ((long)M * N * 4)) : 1;
  if (splits > 1 && (!ws || ws_bytes < need)) return EGN_E_BADARG;
  g.splits = splits;
  g.K = K / splits;
  float* out = C;
  if (splits > 1) { g.C = (float*)ws; g.ldc = N; }
  int rc;
  switch (variant) {
    case 1: rc = gemm_launch<false, false, 128, 128, 2, 2, 2>(g, st); break;
    default: rc = gemm_launch<false, false, 128, 128, 2, 2, 3>(g, st); break;
  }
  if (rc || splits == 1) return rc;
  const size_t total4 = (size_t)M * N / 4;
  const int grid = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)ws, out, M, N, ldc, splits);
  return (int)hipGetLastError();
}
