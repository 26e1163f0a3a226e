// gemm_dma.hip — experiment: fp32 MFMA GEMM whose LDS tiles are filled by global_load_lds (async DMA, no staging VGPRs, no
// ds_write instructions) and whose A fragments come from one ds_read_b128 per 4 MFMAs + v_permlane32_swap (k stays ascending, so the
// result is the same k-ordered fmaf chain as igemm.h).  C[M][N] = relu(A[M][K] * B[K][N] + bias).  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ void glds16(const float* g_lane, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                   0, 0);
}

// 64x64 tile, BR = 32, 4 waves as 2x2, one 32x32 accumulator per wave
__global__ __launch_bounds__(256, 4) void gemm_dma_kernel(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, float* dbg = nullptr) {
  constexpr int BX = 64, BY = 64, BR = 32;
  constexpr int ASZ = (BR / 4) * BX * 4, BSZ = BR * BY;   // floats
  __shared__ __attribute__((aligned(16))) float smem[2 * (ASZ + BSZ)];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave >> 1, wy = wave & 1;
  const int x0 = blockIdx.x * BX, y0 = blockIdx.y * BY;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  // per-lane global row for the A DMA (lane = x), clamped
  const float* arow = A + (size_t)min(x0 + lane, M - 1) * K;
  auto dma = [&](int r0, int buf) {
    float* As = smem + buf * (ASZ + BSZ);
    float* Bs = As + ASZ;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int rq = 2 * wave + q;                                       // A[rq][x][4]: this instruction fills x = 0..63 of one rq
      glds16(arow + r0 + 4 * rq, As + (rq * BX) * 4);
      const int u0 = (2 * wave + q) * 64;                                // B[k][n]: 64 consecutive 16-byte units = 4 k rows
      const int u = u0 + lane, k = u >> 4, n4 = u & 15;
      glds16(B + (size_t)(r0 + k) * N + y0 + 4 * n4, Bs + u0 * 4);
    }
  };
  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (dbg && blockIdx.x == 0 && blockIdx.y == 0) for (int i = tid; i < ASZ + BSZ; i += 256) dbg[i] = smem[i];
  int buf = 0;
  for (int r0 = 0; r0 < K; r0 += BR) {
    if (r0 + BR < K) dma(r0 + BR, buf ^ 1);
    const float* As = smem + buf * (ASZ + BSZ);
    const float* Bs = As + ASZ;
#pragma unroll
    for (int g = 0; g < BR / 8; ++g) {
#ifdef NO_SWAP
      const float* ax = As + ((2 * g) * BX + wx * 32 + li) * 4;
      const float a0 = ax[h], a1 = ax[2 + h], a2 = ax[BX * 4 + h], a3 = ax[BX * 4 + 2 + h];
#else
      float4 av = *reinterpret_cast<const float4*>(As + (((2 * g + h) * BX) + wx * 32 + li) * 4);
      unsigned r0u = __builtin_bit_cast(unsigned, av.x), r1u = __builtin_bit_cast(unsigned, av.y);
      unsigned r2u = __builtin_bit_cast(unsigned, av.z), r3u = __builtin_bit_cast(unsigned, av.w);
      // (the __builtin_amdgcn_permlane32_swap of this toolchain returns a wrong second element: use the instruction directly)
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r0u), "+v"(r1u));   // r0u = (k0 | k1), r1u = (k4 | k5)
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r2u), "+v"(r3u));   // r2u = (k2 | k3), r3u = (k6 | k7)
      const float a0 = __builtin_bit_cast(float, r0u), a2 = __builtin_bit_cast(float, r1u);
      const float a1 = __builtin_bit_cast(float, r2u), a3 = __builtin_bit_cast(float, r3u);
#endif
      const float* bp = Bs + (8 * g + h) * BY + wy * 32 + li;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bp[0 * BY], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bp[2 * BY], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, bp[4 * BY], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, bp[6 * BY], acc, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)");
    __syncthreads();
    buf ^= 1;
  }
  const int n = y0 + wy * 32 + li;
  for (int e = 0; e < 16; ++e) {
    const int m = x0 + wx * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (m < M) { const float v = acc[e] + bias[n]; C[(size_t)m * N + n] = v > 0.f ? v : 0.f; }
  }
}

__global__ void ref_kernel(const float* A, const float* B, const float* bias, float* C, int M, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float acc = 0.0f;
  for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)m * K + k], B[(size_t)k * N + n], acc);
  const float v = acc + bias[n];
  C[(size_t)m * N + n] = v > 0.f ? v : 0.f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 3840, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 3136;
  std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hb(N);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) / 1000.0f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) / 4000.0f;
  for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.0f;
  float *A, *B, *b, *C, *R;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&b, N * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&R, (size_t)M * N * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(ref_kernel, dim3(N / 256, M), dim3(256), 0, 0, A, B, b, R, M, N, K);
  dim3 grid((M + 63) / 64, N / 64);
  hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(256), 0, 0, A, B, b, C, M, N, K);
  hipDeviceSynchronize();
  {
    float* dbg; hipMalloc(&dbg, 4096 * 4);
    hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(256), 0, 0, A, B, b, C, M, N, K, dbg);
    hipDeviceSynchronize();
    std::vector<float> hd(4096); hipMemcpy(hd.data(), dbg, 4096 * 4, hipMemcpyDeviceToHost);
    size_t badA = 0, badB = 0;
    for (int rq = 0; rq < 8; ++rq) for (int x = 0; x < 64; ++x) for (int kk = 0; kk < 4; ++kk)
      if (hd[(rq * 64 + x) * 4 + kk] != hA[(size_t)x * K + 4 * rq + kk]) ++badA;
    for (int k = 0; k < 32; ++k) for (int n = 0; n < 64; ++n) if (hd[2048 + k * 64 + n] != hB[(size_t)k * N + n]) ++badB;
    printf("LDS tile check: A mismatches %zu / 2048, B mismatches %zu / 2048\n", badA, badB);
    if (badA) { printf("A row0 rq0: got %g %g %g %g want %g %g %g %g ; LDS[256..259] = %g %g %g %g\n", hd[0], hd[1], hd[2], hd[3], hA[0], hA[1], hA[2], hA[3], hd[256], hd[257], hd[258], hd[259]); }
  }
  std::vector<float> hC((size_t)M * N), hR((size_t)M * N);
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0; double maxd = 0;
  for (size_t i = 0; i < hC.size(); ++i) { if (hC[i] != hR[i]) ++bad; double d = fabs((double)hC[i] - hR[i]); if (d > maxd) maxd = d; }
  printf("mismatching elements (bitwise): %zu of %zu, max abs diff %g\n", bad, hC.size(), maxd);
  if (bad && M * N <= 64 * 64) {
    for (int m = 0; m < M; m += 1) { for (int n = 0; n < N; n += 1) putchar(hC[(size_t)m * N + n] == hR[(size_t)m * N + n] ? '.' : 'X'); putchar('\n'); }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(256), 0, 0, A, B, b, C, M, N, K);
  hipEventRecord(e0);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(256), 0, 0, A, B, b, C, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("gemm_dma 64x64x32: %.1f us  %.1f TFLOP/s\n", us, tf);
  return 0;
}
