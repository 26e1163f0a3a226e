// micro-benchmark: register-fed v_mfma_f32_16x16x4_f32 against v_mfma_f32_32x32x2_f32 (same flops per cycle on paper: 2048 / 32 vs 4096 / 64) with
// 1..4 independent accumulators per wave and 1 or 2 waves per SIMD, plus distinct B registers per step like a weights-in-registers kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NB>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[NACC], b[NB];
  for (int n = 0; n < NACC; ++n) a[n] = a0 + threadIdx.x + n;
  for (int j = 0; j < NB; ++j) b[j] = b0 + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n], b[u % NB], acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) s += acc[n][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int NB>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  float a[NACC], b[NB];
  for (int n = 0; n < NACC; ++n) a[n] = a0 + threadIdx.x + n;
  for (int j = 0; j < NB; ++j) b[j] = b0 + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[n], b[u % NB], acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K>
void run(const char* name, K kern, int blocks, int threads, int iters, double flops_per_wave_iter) {
  float* out; hipMalloc(&out, blocks * threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 8, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * (threads / 64) * iters * flops_per_wave_iter;
  printf("%-44s blocks=%5d x %3d  %8.1f us  %7.1f TF  (%.3f of 157.3)\n", name, blocks, threads, ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
  hipFree(out);
}
int main() {
  run("32x32x2 4 acc, 1 wave/SIMD", k32<4, 16>, 256, 256, 2000, 16 * 4 * 4096.0);
  run("32x32x2 2 acc, 1 wave/SIMD", k32<2, 16>, 256, 256, 4000, 16 * 2 * 4096.0);
  run("32x32x2 2 acc, 2 waves/SIMD", k32<2, 16>, 512, 256, 2000, 16 * 2 * 4096.0);
  run("32x32x2 1 acc, 2 waves/SIMD", k32<1, 16>, 512, 256, 4000, 16 * 1 * 4096.0);
  run("16x16x4 4 acc, 1 wave/SIMD", k16<4, 32>, 256, 256, 2000, 32 * 4 * 2048.0);
  run("16x16x4 2 acc, 1 wave/SIMD", k16<2, 32>, 256, 256, 4000, 32 * 2 * 2048.0);
  run("16x16x4 2 acc, 2 waves/SIMD", k16<2, 32>, 512, 256, 2000, 32 * 2 * 2048.0);
  run("16x16x4 2 acc, 2 waves/SIMD (512-thread blocks)", k16<2, 32>, 256, 512, 2000, 32 * 2 * 2048.0);
  run("16x16x4 1 acc, 2 waves/SIMD", k16<1, 32>, 512, 256, 4000, 32 * 1 * 2048.0);
  run("16x16x4 1 acc, 4 waves/SIMD", k16<1, 32>, 1024, 256, 2000, 32 * 1 * 2048.0);
  run("16x16x4 4 acc, 2 waves/SIMD", k16<4, 32>, 512, 256, 1000, 32 * 4 * 2048.0);
  return 0;
}
