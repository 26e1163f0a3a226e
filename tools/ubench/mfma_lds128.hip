// micro-benchmark: MFMA f32 fed by ds_read_b128 (one read feeds 4 MFMAs per operand) vs ds_read_b32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MODE>  // MODE 0: b32 reads, 1: b128 reads (pitch 36), 2: b64
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
  __shared__ __attribute__((aligned(16))) float sm[2 * 128 * 36];
  for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) sm[i] = a0 + i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
  const float* A = sm + (w * 32 + li) * 36 + 4 * h;
  const float* B = sm + 128 * 36 + (li) * 36 + 4 * h;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {   // 4 groups of 8 k = 32 k per "chunk" = 16 MFMAs per acc
      if (MODE == 1) {
        const float4 a = *reinterpret_cast<const float4*>(A + 8 * ((g + it) & 3));
        const float4 b = *reinterpret_cast<const float4*>(B + 8 * ((g + it) & 3));
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[n], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a = A[8 * ((g + it) & 3) + q], b = B[8 * ((g + it) & 3) + q];
#pragma unroll
          for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
        }
      }
    }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int MODE>
void run(const char* name, int blocks, int iters) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, 8, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
  printf("%-28s blocks=%5d iters=%4d  %8.1f us  %7.1f TF\n", name, blocks, iters, ms * 1e3, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1, 0>("b32 1acc 4blk/CU", 1024, 500);
  run<1, 1>("b128 1acc 4blk/CU", 1024, 500);
  run<2, 0>("b32 2acc 4blk/CU", 1024, 250);
  run<2, 1>("b128 2acc 4blk/CU", 1024, 250);
  run<1, 1>("b128 1acc 2blk/CU", 512, 1000);
  run<1, 1>("b128 1acc 1blk/CU", 256, 2000);
  run<1, 1>("b128 1acc short 16", 6000, 16);
  return 0;
}
