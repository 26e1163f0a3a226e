// micro-benchmark: v_mfma_f32_32x32x2_f32 rate under different issue patterns (calibrates the f32 MFMA roofline)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = a0 + i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (LDS) { a = sm[(threadIdx.x + u * 64 + it) & 4095]; b = sm[(threadIdx.x * 33 + u + it) & 4095]; }
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool LDS>
void run(const char* name, int blocks, int iters) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, 8, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
  printf("%-28s blocks=%5d iters=%4d  %8.1f us  %7.1f TF\n", name, blocks, iters, ms * 1e3, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4, false>("4acc reg 1blk/CU long", 256, 4000);
  run<8, false>("8acc reg 1blk/CU long", 256, 2000);
  run<8, false>("8acc reg 2blk/CU long", 512, 2000);
  run<8, true>("8acc lds 2blk/CU long", 512, 2000);
  run<4, true>("4acc lds 4blk/CU long", 1024, 2000);
  run<1, false>("1acc reg", 256, 2000);
  run<1, false>("1acc reg 2blk/CU", 512, 1000);
  run<1, false>("1acc reg 4blk/CU", 1024, 500);
  run<2, false>("2acc reg", 256, 1000);
  run<1, true>("1acc lds", 256, 2000);
  run<1, true>("1acc lds 4blk/CU", 1024, 500);
  run<2, true>("2acc lds 4blk/CU", 1024, 250);
  run<1, true>("1acc lds short blocks", 12000, 8);
  run<1, true>("1acc lds short blocks16", 6000, 16);
  run<2, true>("2acc lds short blocks16", 3000, 16);
  return 0;
}
