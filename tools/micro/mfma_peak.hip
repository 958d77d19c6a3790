// mfma_peak.hip -- what the fp32 MFMA pipe of gfx950 sustains, per instruction shape and
// waves per SIMD (standalone: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o mfma_peak).
// Every wave runs ITER iterations of NACC independent MFMAs (no memory traffic, operands in
// registers), so the only limits are the pipe itself and the issue of one or more waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}

template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.f) out[0] = s;
}

template <typename K>
static void run(const char* name, K kern, int threads, int nacc, double flop_per_mfma, float* out) {
  const int cus = 256, iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, out, 100, 1.f, 2.f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mfmas = (double)cus * (threads / 64) * (double)iters * nacc;
  printf("%-28s waves/SIMD %d  acc %d  %8.3f ms  %7.1f TFLOP/s\n", name, threads / 256, nacc, best,
         mfmas * flop_per_mfma / (best * 1e-3) / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  for (int th : {256, 512, 768, 1024}) {
    run("v_mfma_f32_16x16x4_f32", k16<4>, th, 4, 2048.0, out);
    run("v_mfma_f32_16x16x4_f32", k16<12>, th, 12, 2048.0, out);
    run("v_mfma_f32_32x32x2_f32", k32<2>, th, 2, 4096.0, out);
    run("v_mfma_f32_32x32x2_f32", k32<4>, th, 4, 4096.0, out);
  }
  return 0;
}
