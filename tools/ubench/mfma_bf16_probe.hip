// Numerics probe of v_mfma_f32_32x32x16_bf16 (gfx950): one instruction per wave on given operands, inputs and outputs dumped raw for tools/mfma_bf16_model.py,
// which holds candidate summation models against them (is D = round(C + sum of the 16 exact products) with ONE rounding? in which order / width otherwise?).
//   ./mfma_bf16_probe <cases> <out.bin>     layout per case: A[32][16] bf16 (uint16), B[16][32] bf16, C[32][32] f32, D[32][32] f32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(64) void probe(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
  const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
  const size_t cs = blockIdx.x;
  u16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = A[cs * 512 + li * 16 + 8 * h + j]; b[j] = B[cs * 512 + (8 * h + j) * 32 + li]; }   // A[row li][k = 8h + j], B[k = 8h + j][col li]
  f32x16 c;
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = C[cs * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + li];
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 16; ++e) D[cs * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + li] = c[e];
}

static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 32); }
static float unif() { return (rnd() >> 8) * (1.0f / 16777216.0f); }
static float gauss() { float s = 0; for (int i = 0; i < 12; ++i) s += unif(); return s - 6.0f; }
static uint16_t bf16_trunc(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const char* path = argc > 2 ? argv[2] : "mfma_bf16_probe.bin";
  std::vector<uint16_t> A((size_t)n * 512), B((size_t)n * 512);
  std::vector<float> C((size_t)n * 1024), D((size_t)n * 1024);
  for (int cs = 0; cs < n; ++cs) {
    const int mode = cs % 6;
    for (int i = 0; i < 512; ++i) {
      float a, b;
      switch (mode) {
        case 0: a = (float)(rnd() % 17) - 8.0f; b = (float)(rnd() % 17) - 8.0f; break;                                 // small integers: every sum exact
        case 1: a = gauss(); b = gauss(); break;
        case 2: a = unif() < 0.13f ? (float)(1 + rnd() % 255) : 0.0f; b = gauss() * 0.05f / 255.0f * ((cs / 6) % 3 == 0 ? 1.0f : (cs / 6) % 3 == 1 ? 0.0039f : 1.5e-5f); break;   // conv1: pixels x weight terms
        case 3: a = (rnd() & 1) ? 1.0f : -1.0f; b = ldexpf(1.0f, (int)(rnd() % 12) - 6); break;                         // +-1 x powers of two
        case 4: a = gauss() * ldexpf(1.0f, (int)(rnd() % 30) - 15); b = gauss(); break;                                   // wide exponent spread
        default: a = 1.0f; b = (i % 7 == 0) ? 1.0f : 0.0f; break;                                                        // a few unit products against big accumulators
      }
      A[(size_t)cs * 512 + i] = bf16_trunc(a);
      B[(size_t)cs * 512 + i] = bf16_trunc(b);
    }
    for (int i = 0; i < 1024; ++i) {
      float c;
      switch (mode) {
        case 0: c = (float)(rnd() % 65) - 32.0f; break;
        case 1: c = gauss() * 4.0f; break;
        case 2: c = gauss() * 0.5f; break;
        case 3: c = ldexpf(1.0f + unif(), 20 + (int)(rnd() % 8)) * ((rnd() & 1) ? 1.f : -1.f); break;
        case 4: c = gauss() * ldexpf(1.0f, (int)(rnd() % 30) - 15); break;
        default: c = ldexpf(1.0f, 24) + 2.0f * (float)(rnd() % 8); break;
      }
      C[(size_t)cs * 1024 + i] = c;
    }
  }
  uint16_t *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, dA, dB, dC, dD);
  if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "probe failed\n"); return 1; }
  FILE* f = fopen(path, "wb");
  for (int cs = 0; cs < n; ++cs) {
    fwrite(&A[(size_t)cs * 512], 2, 512, f); fwrite(&B[(size_t)cs * 512], 2, 512, f);
    fwrite(&C[(size_t)cs * 1024], 4, 1024, f); fwrite(&D[(size_t)cs * 1024], 4, 1024, f);
  }
  fclose(f);
  printf("wrote %d cases to %s\n", n, path);
  return 0;
}
