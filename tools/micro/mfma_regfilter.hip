// mfma_regfilter.hip -- would a 48x48x3x3 filter held in REGISTERS pay?  One wave per SIMD keeps all
// 27 x 3 B fragments (324 VGPRs/AGPRs) and reads only the A fragments from LDS: MT rows per wave,
// 12*MT MFMAs per MT ds_read_b128 and K step.  Compare with mfma_lds.hip (4 reads per 12 MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT>
__global__ __launch_bounds__(256, 1) void kern(float* out, const float4* __restrict__ w, int iters) {
  extern __shared__ float4 sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += blockDim.x) sm[i] = make_float4(i * 1e-6f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  float4 bf[27][3];
#pragma unroll
  for (int s = 0; s < 27; ++s)
#pragma unroll
    for (int n = 0; n < 3; ++n) bf[s][n] = w[(s * 3 + n) * 64 + lane];
  f32x4 acc[MT][3];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 af[2][MT];
  int pa = lane * 4;
#define LOADA(K, S)                                                                              \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) af[K][m] = sm[(pa + m * 72 + (S)*4) & 8191];
#define MFMA(K, S)                                                                                     \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int n = 0; n < 3; ++n) {       \
    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][m].x, bf[S][n].x, acc[m][n], 0, 0, 0);      \
    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][m].y, bf[S][n].y, acc[m][n], 0, 0, 0);      \
    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][m].z, bf[S][n].z, acc[m][n], 0, 0, 0);      \
    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[K][m].w, bf[S][n].w, acc[m][n], 0, 0, 0);      \
  }
#define PAIR(S) LOADA(1, (S) + 1) MFMA(0, S) LOADA(0, (S) + 2) MFMA(1, (S) + 1)
  for (int it = 0; it < iters; ++it) {
    LOADA(0, 0)
    PAIR(0) PAIR(2) PAIR(4) PAIR(6) PAIR(8) PAIR(10) PAIR(12) PAIR(14) PAIR(16) PAIR(18) PAIR(20) PAIR(22) PAIR(24)
    MFMA(0, 26)
    pa += 64;
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 3; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  if (s == 12345.f) out[0] = s;
}

template <typename K>
static void run(const char* name, K k, int mt, float* out, float4* w) {
  const int cus = 256, iters = 400, threads = 256;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 131072, 0, out, w, 4);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 131072, 0, out, w, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mfmas = (double)cus * (threads / 64) * (double)iters * 27 * 12 * mt;
  printf("%-40s waves/SIMD 1  %8.3f ms  %7.1f TFLOP/s  (%s)\n", name, best, mfmas * 2048.0 / (best * 1e-3) / 1e12,
         hipGetErrorString(hipGetLastError()));
}

int main() {
  float* out;
  float4* w;
  hipMalloc(&out, 64);
  hipMalloc(&w, 27 * 3 * 64 * 16);
  hipMemset(w, 0, 27 * 3 * 64 * 16);
  run("filter in registers, MT=2", kern<2>, 2, out, w);
  run("filter in registers, MT=4", kern<4>, 4, out, w);
  return 0;
}
