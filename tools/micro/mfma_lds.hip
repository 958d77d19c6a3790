// mfma_lds.hip -- the inner loop of the filter-resident conv kernel in isolation: per K step
// 4 x ds_read_b128 (one A fragment, three B fragments) feed 12 MFMAs; fragments double-buffered
// in registers.  Variants: V=0 plain, V=1 with the MFMA/other sched_group_barrier interleave of
// conv_c48.hip, V=2 s_barrier every 27 steps, V=3 both.  MT=2: two A fragments, 24 MFMAs per 5 reads.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, int MT>
__global__ void kern(float* out, int iters, int stride) {
  extern __shared__ float4 sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += blockDim.x) sm[i] = make_float4(i * 1e-6f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  f32x4 acc[MT][3];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 af[2][MT], bf[2][3];
  int pa = lane * 4, pb = 4096 + lane;
#define LOADF(K)                                                                     \
  {                                                                                  \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) af[K][m] = sm[(pa + m * 72) & 4095]; \
    _Pragma("unroll") for (int n = 0; n < 3; ++n) bf[K][n] = sm[4096 + ((pb + n * 16) & 4095)]; \
    pa += stride;                                                                    \
    pb += 192;                                                                       \
  }
#define MFMA(K)                                                                                        \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int n = 0; n < 3; ++n) {       \
    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][m].x, bf[K][n].x, acc[m][n], 0, 0, 0);      \
    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][m].y, bf[K][n].y, acc[m][n], 0, 0, 0);      \
    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][m].z, bf[K][n].z, acc[m][n], 0, 0, 0);      \
    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][m].w, bf[K][n].w, acc[m][n], 0, 0, 0);      \
  }
#define INTER()                                                                      \
  if (V & 1) {                                                                       \
    _Pragma("unroll") for (int k_ = 0; k_ < MT * 12; ++k_) {                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
      __builtin_amdgcn_sched_group_barrier(0x106, 1, 0);                             \
    }                                                                                \
  }
  LOADF(0)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      LOADF(1) MFMA(0) INTER() LOADF(0) MFMA(1) INTER()
    }
    if (V & 2) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 3; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  if (s == 12345.f) out[0] = s;
}

template <typename K>
static void run(const char* name, K k, int threads, int mt, float* out) {
  const int cus = 256, iters = 400;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 131072, 0, out, 4, 4);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 131072, 0, out, iters, 4);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mfmas = (double)cus * (threads / 64) * (double)iters * 26 * 12 * mt;
  printf("%-44s waves/SIMD %d  %8.3f ms  %7.1f TFLOP/s\n", name, threads / 256, best,
         mfmas * 2048.0 / (best * 1e-3) / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  for (int th : {256, 512, 768}) {
    run("MT=1 plain", kern<0, 1>, th, 1, out);
    run("MT=1 sched_group_barrier interleave", kern<1, 1>, th, 1, out);
    run("MT=1 barrier / 26 steps", kern<2, 1>, th, 1, out);
    run("MT=1 interleave + barrier", kern<3, 1>, th, 1, out);
    run("MT=2 plain", kern<0, 2>, th, 2, out);
    run("MT=2 interleave + barrier", kern<3, 2>, th, 2, out);
  }
  return 0;
}
