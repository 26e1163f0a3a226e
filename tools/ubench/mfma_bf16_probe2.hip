// Structured numerics probe of v_mfma_f32_32x32x16_bf16 (see mfma_bf16_probe.hip for the random one): each case sets output (0,0)'s 16 products and accumulator
// from a text script and prints D[0][0].   ./mfma_bf16_probe2 < script      script line: c a0 b0 a1 b1 ... a15 b15   (floats; a / b must be bf16-representable)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(64) void probe(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
  const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
  const size_t cs = blockIdx.x;
  u16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = li == 0 ? A[cs * 16 + 8 * h + j] : 0; b[j] = li == 0 ? B[cs * 16 + 8 * h + j] : 0; }   // row 0 of A, column 0 of B
  f32x16 c;
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.0f;
  if (lane == 0) c[0] = C[cs];
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  if (lane == 0) D[cs] = c[0];
}
static uint16_t bf(float v) { uint32_t u; memcpy(&u, &v, 4); if (u & 0xffff) fprintf(stderr, "not bf16: %g\n", v); return (uint16_t)(u >> 16); }
int main() {
  std::vector<uint16_t> A, B; std::vector<float> C;
  float c;
  while (scanf("%f", &c) == 1) {
    C.push_back(c);
    for (int k = 0; k < 16; ++k) { float a, b; if (scanf("%f %f", &a, &b) != 2) return 1; A.push_back(bf(a)); B.push_back(bf(b)); }
  }
  const int n = (int)C.size();
  uint16_t *dA, *dB; float *dC, *dD;
  (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, n * 4); (void)hipMalloc(&dD, n * 4);
  (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dC, C.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, dA, dB, dC, dD);
  std::vector<float> D(n);
  if (hipMemcpy(D.data(), dD, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  for (int i = 0; i < n; ++i) { uint32_t u; memcpy(&u, &D[i], 4); printf("%d %.9g 0x%08x\n", i, D[i], u); }
  return 0;
}
