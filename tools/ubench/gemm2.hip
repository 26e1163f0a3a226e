// gemm2.hip — experiment (round 3): fp32-MFMA GEMM core fed ONLY by the load unit.
//   C[M][N] = sum_k A(m,k) * B(k,n), both LDS tiles filled by global_load_lds_dwordx4 (no staging VGPRs, no ds_write), ST-stage ring,
//   one barrier per K chunk, TMxTN accumulators of 32x32 per wave.  Optional: the B operand (weights) bypasses LDS entirely — it is read
//   from a pre-arranged copy straight into MFMA operand registers (one dwordx4 per lane = 4 k-steps of one 32-column tile).
// Operand storage modes:  AK = 1: A is [M][K] (k contiguous, LDS [k/4][BM][4])   AK = 0: A is [K][M] (m contiguous, LDS [k][BM])
//                         BKM = 1: B is [N][K] (k contiguous)                     BKM = 0: B is [K][N]
// The K loop is k-ascending per accumulator: bit-identical to an fmaf chain (checked against ref_kernel).
// Build: hipcc -O3 --offload-arch=gfx950 gemm2.hip -o gemm2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ void glds16(const float* g_lane, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                   0, 0);
}
#ifndef ABL
#define ABL 0
#endif
template <int N> static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int BKC, int WM, int WN, int AK, int BKM, int ST, int BD, int MINW>
__global__ __launch_bounds__(WM* WN * 64, MINW) void gemm2_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Bp,
                                                                 float* __restrict__ C, int M, int N, int K, int kslice, unsigned long long* clk, unsigned long long* trace) {
  const unsigned long long t_start = __builtin_readcyclecounter();
  constexpr int NW = WM * WN, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int ASZ = BM * BKC, BSZ = BD ? 0 : BN * BKC;
  constexpr int NA = ASZ / 256, NB = BSZ / 256;        // DMA instructions per chunk (1 KiB each)
  static_assert(NA % NW == 0 && NB % NW == 0, "DMA instructions must divide evenly over the waves");
  constexpr int DA = NA / NW, DB = NB / NW, DPW = DA + DB;
  __shared__ __attribute__((aligned(16))) float smem[ST * (ASZ + BSZ)];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave / WN, wy = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const int k_lo = z * kslice, k_hi = min(K, k_lo + kslice), nc = (k_hi - k_lo) / BKC;

  // per-lane element offsets (floats) of this wave's DMA instructions inside a chunk, relative to the chunk origin
  uint32_t aoff[DA], boff[DB > 0 ? DB : 1];
#pragma unroll
  for (int t = 0; t < DA; ++t) {
    const int j = wave + NW * t;
    if (AK) { const int kq = j / (BM / 64), rb = j % (BM / 64); aoff[t] = (uint32_t)min(m0 + rb * 64 + lane, M - 1) * (uint32_t)K + 4 * kq; }
    else { const int e = j * 256 + lane * 4, k = e / BM, m = e % BM; aoff[t] = (uint32_t)k * (uint32_t)M + min(m0 + m, M - 4); }
  }
#pragma unroll
  for (int t = 0; t < DB; ++t) {
    const int j = wave + NW * t;
    if (BKM) { const int kq = j / (BN / 64), rb = j % (BN / 64); boff[t] = (uint32_t)min(n0 + rb * 64 + lane, N - 1) * (uint32_t)K + 4 * kq; }
    else { const int e = j * 256 + lane * 4, k = e / BN, n = e % BN; boff[t] = (uint32_t)k * (uint32_t)N + min(n0 + n, N - 4); }
  }
  auto dma = [&](int c) __attribute__((always_inline)) {
    float* As = smem + (c % ST) * (ASZ + BSZ);
    float* Bs = As + ASZ;
    const int k0 = k_lo + c * BKC;
    const float* Ac = A + (AK ? (size_t)k0 : (size_t)k0 * M);   // wave-uniform chunk origin
    const float* Bc = B + (BKM ? (size_t)k0 : (size_t)k0 * N);
#pragma unroll
    for (int t = 0; t < DA; ++t) glds16(Ac + aoff[t], As + (wave + NW * t) * 256);
#pragma unroll
    for (int t = 0; t < DB; ++t) glds16(Bc + boff[t], Bs + (wave + NW * t) * 256);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // direct-B: Bp[(tile32 * (K/8) + g) * 256 + lane * 4 + q] = B(k = 8g + 2q + h, n = 32*tile + li)
  constexpr int NG = BKC / 8;                 // dwordx4 loads per tile per chunk
  f32x4 bq[2][TN][NG];
  auto bload = [&](int c, int set) __attribute__((always_inline)) {
    if constexpr (BD) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int tile = (n0 >> 5) + wy * TN + j;
          bq[set][j][g] = *reinterpret_cast<const f32x4*>(Bp + ((size_t)tile * (K / 8) + (k_lo + c * BKC) / 8 + g) * 256 + lane * 4);
        }
    }
  };

  auto compute = [&](int c, int set) __attribute__((always_inline)) {
    const float* As = smem + (c % ST) * (ASZ + BSZ);
    const float* Bs = As + ASZ;
    constexpr int G = 4, NGR = BKC / 2 / G;
    float fa[2][G][TM], fb[2][G][TN];
    auto frag = [&](int g, int fs) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < G; ++q) {
        const int s = g * G + q;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int x = wx * (TM * 32) + i * 32 + li;
          fa[fs][q][i] = AK ? As[((s >> 1) * BM + x) * 4 + 2 * (s & 1) + h] : As[(2 * s + h) * BM + x];
        }
        if constexpr (!BD) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int y = wy * (TN * 32) + j * 32 + li;
            fb[fs][q][j] = BKM ? Bs[((s >> 1) * BN + y) * 4 + 2 * (s & 1) + h] : Bs[(2 * s + h) * BN + y];
          }
        }
      }
    };
    frag(0, 0);
#pragma unroll
    for (int g = 0; g < NGR; ++g) {
      if (g + 1 < NGR && !(ABL & 1)) frag(g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float b = BD ? bq[set][j][g][q] : fb[g & 1][q][j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[(ABL & 1) ? 0 : (g & 1)][q][i], (ABL & 1) && !BD ? fb[0][q][j] : b, acc[i][j], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- pipeline: chunks 0 .. ST-2 in flight before the loop; iteration c: [B regs of c+1], DMA of chunk c+ST-1, multiply chunk c
  bload(0, 0);
#pragma unroll
  for (int s = 0; s < ST - 1; ++s) if (s < nc) dma(s);
  auto iter = [&](int c, int set) __attribute__((always_inline)) {
    if (c + ST - 2 < nc - 0 && ST > 2 && c + 1 < nc) wait_vm<(ST - 2) * DPW>(); else wait_vm<0>();
    if (!(ABL & 4)) asm volatile("s_barrier" ::: "memory");   // (not __syncthreads(): its fence drains vmcnt, i.e. every DMA in flight)
    if (c + 1 < nc) bload(c + 1, set ^ 1);
    if (c + ST - 1 < nc && !((ABL & 2) && c > 2)) dma(c + ST - 1);
    compute(c, set);
  };
  int c = 0;
  for (; c + 1 < nc; c += 2) { iter(c, 0); iter(c + 1, 1); }
  if (c < nc) iter(c, 0);

  if (clk && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) clk[0] = __builtin_readcyclecounter() - t_start;
  if (trace && threadIdx.x == 0) {
    const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    trace[3 * b] = t_start; trace[3 * b + 1] = __builtin_readcyclecounter(); trace[3 * b + 2] = ((unsigned long long)xcc << 32) | hw;
  }
  float* Cz = C + (size_t)z * M * N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wy * (TN * 32) + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wx * (TM * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < M && n < N) Cz[(size_t)m * N + n] = acc[i][j][e];
      }
    }
}

// reference: k-ascending fmaf chain per output, per K slice
template <int AK, int BKM>
__global__ void ref_kernel(const float* A, const float* B, float* C, int M, int N, int K, int kslice) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y, z = blockIdx.z;
  if (n >= N) return;
  float acc = 0.0f;
  for (int k = z * kslice; k < min(K, (z + 1) * kslice); ++k)
    acc = fmaf(AK ? A[(size_t)m * K + k] : A[(size_t)k * M + m], BKM ? B[(size_t)n * K + k] : B[(size_t)k * N + n], acc);
  C[((size_t)z * M + m) * N + n] = acc;
}
template <int BKM>
__global__ void prearrange_kernel(const float* B, float* Bp, int N, int K) {   // one thread per (tile, g, lane)
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = t & 63;
  const size_t r = t >> 6;
  const int g = r % (K / 8), tile = r / (K / 8);
  if (tile >= (N + 31) / 32) return;
  const int li = lane & 31, h = lane >> 5, n = min(tile * 32 + li, N - 1);
  for (int q = 0; q < 4; ++q) { const int k = 8 * g + 2 * q + h; Bp[t * 4 + q] = BKM ? B[(size_t)n * K + k] : B[(size_t)k * N + n]; }
}

static unsigned long long* g_clk = nullptr;
static unsigned long long* g_trace = nullptr;
static const char* g_trace_name = nullptr;
struct Mats { float *A, *B, *Bp, *C, *R; int M, N, K; };
static Mats make(int M, int N, int K, int nz) {
  Mats m{nullptr, nullptr, nullptr, nullptr, nullptr, M, N, K};
  std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) / 1000.0f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) / 4000.0f;
  hipMalloc(&m.A, hA.size() * 4); hipMalloc(&m.B, hB.size() * 4); hipMalloc(&m.Bp, (size_t)((N + 31) / 32) * 32 * K * 4);
  hipMalloc(&m.C, (size_t)nz * M * N * 4); hipMalloc(&m.R, (size_t)nz * M * N * 4);
  hipMemcpy(m.A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(m.B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  return m;
}
static void release(Mats& m) { hipFree(m.A); hipFree(m.B); hipFree(m.Bp); hipFree(m.C); hipFree(m.R); }

template <int BM, int BN, int BKC, int WM, int WN, int AK, int BKM, int ST, int BD, int MINW>
static void run(const char* name, Mats& m, int nz, bool check) {
  const int M = m.M, N = m.N, K = m.K, kslice = ((K / nz + BKC - 1) / BKC) * BKC;
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, nz), blk(WM * WN * 64);
  auto kern = gemm2_kernel<BM, BN, BKC, WM, WN, AK, BKM, ST, BD, MINW>;
  if (BD) hipLaunchKernelGGL(prearrange_kernel<BKM>, dim3((unsigned)(((size_t)((N + 31) / 32) * (K / 8) * 64 + 255) / 256)), dim3(256), 0, 0, m.B, m.Bp, N, K);
  hipMemset(m.C, 0, (size_t)nz * M * N * 4);
  hipLaunchKernelGGL(kern, grid, blk, 0, 0, m.A, m.B, m.Bp, m.C, M, N, K, kslice, g_clk, g_trace);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%-44s LAUNCH FAILED: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
  size_t bad = 0;
  if (check) {
    hipLaunchKernelGGL((ref_kernel<AK, BKM>), dim3((N + 255) / 256, M, nz), dim3(256), 0, 0, m.A, m.B, m.R, M, N, K, kslice);
    std::vector<float> hC((size_t)nz * M * N), hR((size_t)nz * M * N);
    hipMemcpy(hC.data(), m.C, hC.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hR.data(), m.R, hR.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < hC.size(); ++i) if (hC[i] != hR[i]) ++bad;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, blk, 0, 0, m.A, m.B, m.Bp, m.C, M, N, K, kslice, g_clk, g_trace);
  hipEventRecord(e0);
  const int reps = 30;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, blk, 0, 0, m.A, m.B, m.Bp, m.C, M, N, K, kslice, g_clk, g_trace);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  if (g_trace && strstr(name, g_trace_name)) {
    const int nb = grid.x * grid.y * grid.z;
    std::vector<unsigned long long> t(3 * nb);
    hipMemcpy(t.data(), g_trace, t.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < nb; ++b) t0 = t[3 * b] < t0 ? t[3 * b] : t0;
    printf("TRACE %s nb=%d\n", name, nb);
    for (int b = 0; b < nb; ++b) printf("T %d %llu %llu %llx\n", b, t[3 * b] - t0, t[3 * b + 1] - t0, t[3 * b + 2]);
  }
  unsigned long long hclk = 0; hipMemcpy(&hclk, g_clk, 8, hipMemcpyDeviceToHost);
  printf("[blk0 %6.1f kcyc] ", hclk / 1e3);
  printf("%-44s grid %4dx%2dx%2d  %7.1f us  %6.1f TF  %.3f of 157.3   %s\n", name, grid.x, grid.y, grid.z, us, tf, tf / 157.3,
         check ? (bad ? "MISMATCH" : "bit-exact") : "");
  if (bad) printf("   mismatching elements: %zu\n", bad);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipMalloc(&g_clk, 8);
  if (argc > 2) { g_trace_name = argv[2]; hipMalloc(&g_trace, 3 * 8 * 65536); }
  const bool chk = argc > 1 ? atoi(argv[1]) != 0 : true;
  {  // dense forward: hid = act3 [3840][3136] x W [3136][512]
    Mats m = make(3840, 512, 3136, 2);
    printf("== dense fwd 3840 x 512 x 3136  (A k-contiguous, B n-contiguous)\n");
    run<128, 128, 16, 2, 2, 1, 0, 2, 0, 2>("128x128x16 2x2 st2", m, 1, chk);
    run<128, 128, 16, 2, 2, 1, 0, 3, 0, 2>("128x128x16 2x2 st3", m, 1, chk);
    run<128, 128, 32, 2, 2, 1, 0, 2, 0, 2>("128x128x32 2x2 st2", m, 1, chk);
    run<128, 64, 16, 2, 2, 1, 0, 3, 0, 2>("128x64x16 2x2 st3 (wave 64x32)", m, 1, chk);
    run<128, 64, 32, 2, 2, 1, 0, 2, 0, 2>("128x64x32 2x2 st2", m, 1, chk);
    run<64, 128, 16, 2, 2, 1, 0, 3, 0, 2>("64x128x16 2x2 st3 (wave 32x64)", m, 1, chk);
    run<64, 64, 32, 2, 2, 1, 0, 3, 0, 4>("64x64x32 2x2 st3 (1 acc)", m, 1, chk);
    run<256, 128, 16, 4, 2, 1, 0, 2, 0, 2>("256x128x16 4x2 (8 waves) st2", m, 1, chk);
    run<128, 128, 16, 2, 2, 1, 0, 2, 1, 2>("128x128x16 2x2 st2 B-direct", m, 1, chk);
    run<128, 128, 16, 2, 2, 1, 0, 3, 1, 2>("128x128x16 2x2 st3 B-direct", m, 1, chk);
    run<128, 64, 16, 2, 2, 1, 0, 3, 1, 2>("128x64x16 2x2 st3 B-direct", m, 1, chk);
    run<128, 64, 16, 4, 1, 1, 0, 3, 1, 2>("128x64x16 4x1 st3 B-direct (wave 32x64)", m, 1, chk);
    run<128, 128, 16, 2, 2, 1, 0, 3, 0, 2>("128x128x16 2x2 st3 split-K 2", m, 2, chk);
    run<128, 128, 16, 2, 2, 1, 0, 3, 1, 2>("128x128x16 2x2 st3 B-direct split-K 2", m, 2, chk);
    release(m);
  }
  {  // dense dgrad: dact3 = dhid [3840][512] x W^T  (W stored [3136][512] = B k-contiguous), N padded to 3200 for the timing
    Mats m = make(3840, 3200, 512, 1);
    printf("== dense dgrad 3840 x 3200 x 512  (A k-contiguous, B k-contiguous)\n");
    run<128, 128, 16, 2, 2, 1, 1, 2, 0, 2>("128x128x16 2x2 st2", m, 1, chk);
    run<128, 128, 16, 2, 2, 1, 1, 3, 0, 2>("128x128x16 2x2 st3", m, 1, chk);
    run<128, 128, 32, 2, 2, 1, 1, 2, 0, 2>("128x128x32 2x2 st2", m, 1, chk);
    run<256, 128, 16, 4, 2, 1, 1, 2, 0, 2>("256x128x16 4x2 (8 waves) st2", m, 1, chk);
    run<128, 128, 16, 2, 2, 1, 1, 3, 1, 2>("128x128x16 2x2 st3 B-direct", m, 1, chk);
    run<128, 256, 16, 2, 2, 1, 1, 3, 1, 1>("128x256x16 2x2 st3 B-direct (2x4 acc)", m, 1, chk);
    release(m);
  }
  {  // dense wgrad: dW [3136][512] = act3^T x dhid, reduction over 3840 frames in 10 slices: A m-contiguous, B n-contiguous
    Mats m = make(3200, 512, 3840, 10);
    printf("== dense wgrad 3200 x 512 x 3840, 10 K-slices  (A m-contiguous, B n-contiguous)\n");
    run<128, 256, 16, 2, 2, 0, 0, 2, 0, 1>("128x256x16 2x2 st2 (2x4 acc)", m, 10, chk);
    run<128, 256, 16, 2, 2, 0, 0, 3, 0, 1>("128x256x16 2x2 st3 (2x4 acc)", m, 10, chk);
    run<128, 128, 16, 2, 2, 0, 0, 3, 0, 2>("128x128x16 2x2 st3 x10", m, 10, chk);
    run<128, 128, 16, 2, 2, 0, 0, 3, 0, 2>("128x128x16 2x2 st3 x5", m, 5, chk);
    run<256, 128, 16, 4, 2, 0, 0, 2, 0, 2>("256x128x16 4x2 (8 waves) st2 x10", m, 10, chk);
    release(m);
  }
  return 0;
}
