// micro-benchmark: does VALU work overlap with v_mfma_f32_32x32x2_f32 — inside one wave, and across the waves of a SIMD?
// Each iteration issues NV dependent-free VALU ops (fma chains on private registers) per MFMA, NACC accumulators per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 + threadIdx.x + i;
  float a = a0, b = a0 + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[(j + n) & 7] = __builtin_fmaf(v[(j + n) & 7], 1.0001f, 0.5f);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int NACC>
void run(int blocks, int iters) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NACC>), dim3(blocks), dim3(256), 0, 0, out, 4, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
  printf("VALU/MFMA=%d acc/wave=%d waves/SIMD=%d  %8.1f us  %7.1f TF\n", NV, NACC, blocks / 256, ms * 1e3, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<0, 3>(256, 2000); run<2, 3>(256, 2000); run<5, 3>(256, 2000); run<8, 3>(256, 2000);
  run<0, 3>(512, 1000); run<2, 3>(512, 1000); run<5, 3>(512, 1000); run<8, 3>(512, 1000);
  run<5, 3>(1024, 500); run<8, 3>(1024, 500);
  run<5, 1>(512, 3000); run<5, 1>(1024, 1500);
  return 0;
}
