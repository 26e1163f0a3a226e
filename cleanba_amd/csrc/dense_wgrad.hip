// dense_wgrad.hip — weight gradient of the 3136 -> 512 dense layer (dW[i][j] = sum_f act3[f][i] * dhid[f][j], ppo:619's backward) fed by the load unit.
//
// Both operands have their fast axis along the OUTPUT index (act3 rows run along i, dhid rows along j) and the reduction runs over frames, so a K chunk
// of either tile is a stack of contiguous row pieces: global_load_lds_dwordx4 copies it into LDS as [k][x] with no staging registers and no ds_write,
// and every fragment read (lane = output index, fixed k) is conflict-free.  128x128x16 tiles, 2x2 accumulators per wave, a three-stage ring so that
// the copies of chunk c+2 are issued while chunk c is multiplied, ONE barrier per chunk and — unlike __syncthreads() — no drain of the copies in
// flight at it.  The reduction is cut into 5 frame slices (25 x 4 x 5 = 500 blocks, ~2 per CU): 128.7 -> ~112 us per 3840-frame minibatch against the
// register-staged 128x256 x 10-slice igemm_kernel it replaces, with half the partial traffic (tools/ubench/gemm2.hip has the variants that were timed).
// Partials part[z][3136][512] / bpart[z][512] are reduced in z order by wgrad_reduce_multi_kernel: deterministic, 1e-5 class like every weight gradient.
#include "cbm_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
static __device__ __forceinline__ void dw_glds16(const float* g_lane, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N> static __device__ __forceinline__ void dw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#define DW_BM 128
#define DW_BN 128
#define DW_BK 16
#define DW_ST 3
// A = act [F][X] (X = 3136 outputs of the gradient's rows), G = dhid [F][Y] (Y = 512); block (bx, by, z): rows [bx*128, +128), columns [by*128, +128),
// frames [z*fslice, min(F, (z+1)*fslice)) — fslice a multiple of 16.  F itself need not be (IMPALA's minibatches are 129 x 30 rows): the last chunk of
// the last slice then starts at F - 16, and the rows it shares with the chunk before it enter the products as zeros.
__global__ __launch_bounds__(256, 2) void dense_wgrad_dma_kernel(const float* __restrict__ A, const float* __restrict__ G, float* __restrict__ part,
                                                                 float* __restrict__ bpart, int F, int X, int Y, int fslice) {
  constexpr int ASZ = DW_BM * DW_BK, BSZ = DW_BN * DW_BK, DA = ASZ / 256 / 4, DB = BSZ / 256 / 4, DPW = DA + DB;   // 1 KiB copies per wave and chunk
  // ONE LDS object: with a second one (round 5 had `__shared__ float bred[256]` for the bias sums) hipcc cannot tell the fragment reads from the copies
  // in flight and puts `s_waitcnt vmcnt(0)` in front of the first ds_read of every chunk — the three-stage ring then waits for the copy it has just
  // issued (tools/isa_audit.py; cdna_hip_programming.md section 5, ".s-level traps" (a)): 130 us instead of the 111 the same loop ran in tools/ubench/gemm2.hip
  __shared__ __attribute__((aligned(16))) float smem[DW_ST * (ASZ + BSZ)];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave >> 1, wy = wave & 1;
  // XCD-aware tile order (1-D grid).  Workgroups go to the eight XCDs round robin by linear id and each XCD has its own L2: the id is turned into "XCD j
  // owns the contiguous run [start_j, start_j + n_j) of the list ordered (slice, row tile, column tile)", so the NY column tiles that multiply the same
  // 128-row panel of `act` run back to back on ONE XCD and the panel is fetched into one L2 once.  (As a 3-D grid the four column tiles of a panel sat 25 ids
  // apart — four different XCDs: 246 MB of fabric traffic per launch for 62.5 MB of operands in round 5's counters.)
  const int NX = (X + DW_BM - 1) / DW_BM, NY = (Y + DW_BN - 1) / DW_BN;
  int g = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = g & 7, k = g >> 3;
    g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int by = g % NY, bx = (g / NY) % NX, z = g / (NY * NX);
  const int m0 = bx * DW_BM, n0 = by * DW_BN;
  const int f_lo = z * fslice, f_hi = min(F, f_lo + fslice), nc = (f_hi - f_lo + DW_BK - 1) / DW_BK;
  // element offsets of this wave's copies inside a chunk: copy j moves 256 consecutive floats of the [16][128] tile = 2 k-rows of 128
  uint32_t aoff[DA], boff[DB];
#pragma unroll
  for (int t = 0; t < DA; ++t) { const int e = (wave + 4 * t) * 256 + lane * 4, k = e / DW_BM, m = e % DW_BM; aoff[t] = (uint32_t)k * (uint32_t)X + min(m0 + m, X - 4); }
#pragma unroll
  for (int t = 0; t < DB; ++t) { const int e = (wave + 4 * t) * 256 + lane * 4, k = e / DW_BN, n = e % DW_BN; boff[t] = (uint32_t)k * (uint32_t)Y + min(n0 + n, Y - 4); }
  auto dma = [&](int c) __attribute__((always_inline)) {
    float* As = smem + (c % DW_ST) * (ASZ + BSZ);
    float* Bs = As + ASZ;
    const int k0 = min(f_lo + c * DW_BK, F - DW_BK);   // (a ragged last chunk re-reads rows of its predecessor: masked in compute)
    const float* Ac = A + (size_t)k0 * X;
    const float* Gc = G + (size_t)k0 * Y;
#pragma unroll
    for (int t = 0; t < DA; ++t) dw_glds16(Ac + aoff[t], As + (wave + 4 * t) * 256);
#pragma unroll
    for (int t = 0; t < DB; ++t) dw_glds16(Gc + boff[t], Bs + (wave + 4 * t) * 256);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  float bsum = 0.0f;
  const bool do_bias = bx == 0;            // the column sums of dhid (bias gradient) come from the B tiles of the first row of blocks
  // skip > 0 only for a ragged last chunk: its first `skip` k rows were already multiplied by the chunk before
  auto compute = [&](int c, int skip) __attribute__((always_inline)) {
    const float* As = smem + (c % DW_ST) * (ASZ + BSZ);
    const float* Bs = As + ASZ;
    float fa[DW_BK / 2][2], fb[DW_BK / 2][2];
#pragma unroll
    for (int s = 0; s < DW_BK / 2; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { const float v = As[(2 * s + h) * DW_BM + wx * 64 + i * 32 + li]; fa[s][i] = 2 * s + h >= skip ? v : 0.0f; }
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[s][j] = Bs[(2 * s + h) * DW_BN + wy * 64 + j * 32 + li];
    }
    if (do_bias) {                                 // thread t: column t % 128, k rows (t / 128) * 8 .. + 8, ascending
      const int n = tid & 127, kh = (tid >> 7) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float v = Bs[(kh + k) * DW_BN + n]; bsum += kh + k >= skip ? v : 0.0f; }
    }
#pragma unroll
    for (int s = 0; s < DW_BK / 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < DW_ST - 1; ++s) if (s < nc) dma(s);
  for (int c = 0; c < nc; ++c) {
    if (c + 1 < nc) dw_wait_vm<DPW>(); else dw_wait_vm<0>();   // chunk c has landed; chunk c+1 may still be in flight
    asm volatile("s_barrier" ::: "memory");                    // (not __syncthreads(): its fence would wait for every copy in flight)
    if (c + DW_ST - 1 < nc) dma(c + DW_ST - 1);                // into the buffer chunk c-1 was multiplied from: every wave is past that
    const int over = f_lo + c * DW_BK + DW_BK - F;             // > 0: the chunk was moved back by this many rows to end at F
    if (over > 0) compute(c, over); else compute(c, 0);
  }
  float* Pz = part + (size_t)z * X * Y;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wy * 64 + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wx * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < X && n < Y) Pz[(size_t)m * Y + n] = acc[i][j][e];
      }
    }
  if (do_bias) {
    float* bred = smem;      // the ring is free: every chunk has been multiplied
    __syncthreads();
    bred[tid] = bsum;
    __syncthreads();
    if (tid < 128 && n0 + tid < Y) bpart[(size_t)z * Y + n0 + tid] = bred[tid] + bred[tid + 128];
  }
}

int dense_wgrad_dma_slices(int F) { return F >= 2048 ? 5 : 0; }   // 0: batch not handled here (small batches stay on igemm_kernel)
void launch_dense_wgrad_dma(const float* act, const float* dhid, float* part, float* bpart, int F, int X, int Y, int nz, hipStream_t st) {
  const int fslice = ((F + nz - 1) / nz + DW_BK - 1) / DW_BK * DW_BK;
  dim3 grid(((X + DW_BM - 1) / DW_BM) * ((Y + DW_BN - 1) / DW_BN) * nz);   // 1-D: the kernel orders the tiles per XCD
  hipLaunchKernelGGL(dense_wgrad_dma_kernel, grid, dim3(256), 0, st, act, dhid, part, bpart, F, X, Y, fslice);
}
