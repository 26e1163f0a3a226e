#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float4* in, float* o) {
  __shared__ float4 sm[128];
  sm[threadIdx.x] = in[threadIdx.x]; sm[64 + threadIdx.x] = in[64 + threadIdx.x];
  __syncthreads();
  const int li = threadIdx.x & 31, h = threadIdx.x >> 5;
  float4 av = sm[h * 64 + li];     // like As[(2g+h)*BX + x]
  unsigned r0u = __builtin_bit_cast(unsigned, av.x), r1u = __builtin_bit_cast(unsigned, av.y);
  unsigned r2u = __builtin_bit_cast(unsigned, av.z), r3u = __builtin_bit_cast(unsigned, av.w);
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r0u), "+v"(r1u));
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r2u), "+v"(r3u));
  o[threadIdx.x] = __builtin_bit_cast(float, r0u); o[64 + threadIdx.x] = __builtin_bit_cast(float, r2u);
  o[128 + threadIdx.x] = __builtin_bit_cast(float, r1u); o[192 + threadIdx.x] = __builtin_bit_cast(float, r3u);
}
int main() {
  float4 hin[128];
  for (int rq = 0; rq < 2; ++rq) for (int x = 0; x < 64; ++x) hin[rq * 64 + x] = make_float4(1000 * x + 4 * rq + 0, 1000 * x + 4 * rq + 1, 1000 * x + 4 * rq + 2, 1000 * x + 4 * rq + 3);
  float4* din; float* d; hipMalloc(&din, sizeof(hin)); hipMalloc(&d, 1024); hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, d);
  float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  // expected: a_j lane (li,h) = 1000*li + 2j + h
  int bad = 0;
  for (int j = 0; j < 4; ++j) for (int l = 0; l < 64; ++l) { float want = 1000 * (l & 31) + 2 * j + (l >> 5); if (h[j * 64 + l] != want) { if (bad < 6) printf("a%d lane %d got %g want %g\n", j, l, h[j * 64 + l], want); ++bad; } }
  printf("bad %d\n", bad);
}
