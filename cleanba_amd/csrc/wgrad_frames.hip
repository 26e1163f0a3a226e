// wgrad_frames.hip — frame-resident weight-gradient kernels for conv2 (4x4 stride 2, 32 -> 64) and conv3 (3x3 stride 1, 64 -> 64).
//
// A conv weight gradient is dW[(kh,kw,ci)][co] = sum over (frame, output position) of act[(s*oh+kh, s*ow+kw), ci] * dY[(oh,ow), co]: the
// OUTPUT is tiny (512x64 / 576x64 = 32 / 36 MFMA tiles) and the reduction runs over every position of every frame.  As an implicit GEMM
// (igemm.h ConvWgrad) each 64x64 output tile was its own block: every tap re-gathered the same activation rows through im2col address
// math (6-8 VALU instructions per MFMA) and every sibling tile re-read the same dY slice (L2-miss traffic 2.6x / 6.8x the operands).
// Here a block keeps the WHOLE output in its accumulators (8-9 tiles of 32x32 per wave) and walks frames: the activation frame and the dY
// frame are copied once, linearly, into LDS by the load unit (global_load_lds_dwordx4, no staging registers), and every tap's A fragment
// is the same LDS slab read at a constant offset — the K loop has no address arithmetic at all (two base registers + immediates), one
// barrier pair per frame instead of one per 16-wide K chunk, and HBM / L2 see each operand byte exactly once.  Same structure as
// conv1_wgrad_frames_kernel (conv1.hip), which runs at 0.71 of the fp32 MFMA peak.
//
// MFMA roles (v_mfma_f32_32x32x2_f32, D[x][y] += A[x][k] B[k][y]): x = ci within a tap (lane li), k = output position (2 per instruction:
// lane half h takes position 2q + h), y = co.  Weight gradients carry a 1e-5 bar, not bits: the summation order (positions ascending
// within a frame, frames ascending within a block, blocks reduced in order by wgrad_reduce_multi_kernel) is fixed and deterministic.
#include "cbm_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// (hidden from the compiler: with the builtin, hipcc waited vmcnt(0) for the copy of frame s+1 in front of the first fragment read of frame s —
// cbm_internal.h cbm_glds16_hidden, tools/isa_audit.py)
static __device__ __forceinline__ void glds16(const void* g_lane, void* lds_wave_base) { cbm_glds16_hidden(g_lane, cbm_lds_addr(lds_wave_base)); }
// linear copy of `pieces` 16-byte pieces global -> LDS with all 4 waves (wave-uniform LDS base + lane * 16)
template <int PIECES>
static __device__ __forceinline__ void slab_to_lds(const float* src, float* dst, int wave, int lane) {
  constexpr int ROUNDS = (PIECES + 255) / 256;
#pragma unroll 1   // rolled: unrolled, hipcc keeps every round's 64-bit lane address live across the K loop (and spills accumulators)
  for (int j = 0; j < ROUNDS; ++j) {
    const int v0 = (wave + 4 * j) * 64;
    if (v0 + lane < PIECES) glds16(reinterpret_cast<const char*>(src) + (unsigned)(v0 + lane) * 16u, reinterpret_cast<char*>(dst) + (unsigned)v0 * 16u);
  }
}

// ------------------------------------------------------------------------------------------ conv2: act1 [20][20][32] x dY2 [9][9][64]
// LDS: two stages of {act1 frame 51,200 B + the 9x9 interior of the zero-bordered dY frame 20,736 B} = 143,872 B -> one block per CU, whose
// load unit copies frame s+1 while the four waves multiply frame s (single-buffered, two blocks per CU, every block loaded and computed
// in lock step with all the others — HBM idle during the MFMA phases, MFMAs idle during the load bursts: 229 us).
// wave w owns kernel row kh = w: taps (w, 0..3) x both co halves = 8 independent accumulator tiles.  co is split even / odd (y-tile j
// holds co = 2*li + j) so that ONE ds_read_b64 per lane fetches both B fragments and the partial rows are stored as float2.
#define C2W_A_FLOATS 12800
#define C2W_B_FLOATS 5184
#define C2W_STAGE (C2W_A_FLOATS + C2W_B_FLOATS)
__global__ __launch_bounds__(256, 1) void conv2_wgrad_frames_kernel(const float* act1, const float* dypad, float* part, float* bpart, int S,
                                                                    int frames_per_block) {
  __shared__ __attribute__((aligned(16))) float smem[2 * C2W_STAGE];   // stage: [ih*20 + iw][ci] | [oh*9 + ow][co]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s_lo = blockIdx.x * frames_per_block, s_hi = min(S, s_lo + frames_per_block);
  f32x16 acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.0f;
  float bs0 = 0.0f, bs1 = 0.0f;
  auto stage = [&](int buf, int s) __attribute__((always_inline)) {
    float* As = smem + buf * C2W_STAGE;
    float* Bs = As + C2W_A_FLOATS;
    slab_to_lds<C2W_A_FLOATS / 4>(act1 + (size_t)s * C2W_A_FLOATS, As, wave, lane);
    const float* src = dypad + (size_t)s * 7744;   // dY interior: 9 rows of 9*64 floats (144 pieces each) of the [11][11][64] frame
#pragma unroll 1
    for (int j = 0; j < 6; ++j) {   // 1296 pieces
      const int v0 = (wave + 4 * j) * 64, v = v0 + lane;
      if (v < 1296) {
        const int row = v / 144, col = v - row * 144;
        glds16(reinterpret_cast<const char*>(src) + (unsigned)((((1 + row) * 11 + 1) * 64) * 4 + col * 16), reinterpret_cast<char*>(Bs) + (unsigned)v0 * 16u);
      }
    }
  };
  if (s_lo < s_hi) stage(0, s_lo);
  for (int s = s_lo; s < s_hi; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // frame s has landed for every wave, and every wave is done with the other stage (frame s - 1)
    if (s + 1 < s_hi) stage((s + 1 - s_lo) & 1, s + 1);
    const float* As = smem + ((s - s_lo) & 1) * C2W_STAGE;
    const float* Bs = As + C2W_A_FLOATS;
    // position p = 2q + h.  Offsets of position p: A  (2*oh*20 + 2*ow)*32 floats, B  p*64 floats.  For h = 0 they are compile-time constants
    // (immediates); lane half h = 1 adds the step to position p + 1: +64 floats normally (ow + 1), or to the start of the next output row
    // when ow = 8 -> two base pointers per operand, chosen at compile time per step.
    const float* a_norm = As + (wave * 20) * 32 + li + h * 64;                       // next ow: +2 pixels * 32 channels
    const float* a_wrap = As + (wave * 20) * 32 + li + h * ((2 * 20 - 16) * 32);     // ow = 8 -> (oh + 1, 0)
    const float* a_last = As + (wave * 20) * 32 + li;                                // step 40: both halves read position 80 (h = 1 gets B = 0)
    const float* b_lane = Bs + 2 * li + h * 64;
    // 81 positions = 40 pairs + 1 single (the h = 1 half of the last step multiplies by B = 0)
    float fa[2][4];
    float2 fb[2];
    auto frag = [&](int q, int set) __attribute__((always_inline)) {
      const int p0 = 2 * q, oh = p0 / 9, ow = p0 - oh * 9;
      const int aoff = (2 * oh * 20 + 2 * ow) * 32;
      const float* ap = (q == 40 ? a_last : (ow == 8 ? a_wrap : a_norm)) + aoff;
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[set][t] = ap[t * 32];
      if (q < 40) fb[set] = *reinterpret_cast<const float2*>(b_lane + p0 * 64);
      else { const float2 v = *reinterpret_cast<const float2*>(Bs + 2 * li + 80 * 64); fb[set] = h ? make_float2(0.0f, 0.0f) : v; }
    };
    frag(0, 0);
#pragma unroll
    for (int q = 0; q < 41; ++q) {
      if (q + 1 < 41) frag(q + 1, (q + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const float2 b = fb[q & 1];
      if (wave == 0) { bs0 += b.x; bs1 += b.y; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][t], b.x, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][t], b.y, acc[t][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // partial out: part[block][(kh*4 + kw)*32 + ci][co], ci = tile row, co = 2*li + j
  float* o = part + (size_t)blockIdx.x * 512 * 64;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      *reinterpret_cast<float2*>(o + (size_t)((wave * 4 + t) * 32 + row) * 64 + 2 * li) = make_float2(acc[t][0][e], acc[t][1][e]);
    }
  if (wave == 0) {
    bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 32, 64);
    if (h == 0) *reinterpret_cast<float2*>(bpart + (size_t)blockIdx.x * 64 + 2 * li) = make_float2(bs0, bs1);
  }
}

static const int C2W_BLOCKS = 256;   // one block per CU
int conv2_wgrad_frames_splits(int S) {
  int blocks = C2W_BLOCKS;
  if (S < blocks) blocks = S;
  const int fpb = (S + blocks - 1) / blocks;
  return (S + fpb - 1) / fpb;
}
int conv2_wgrad_frames_splits_bound(int maxS) { return maxS < C2W_BLOCKS ? maxS : C2W_BLOCKS; }
void launch_conv2_wgrad_frames(const float* act1, const float* dypad, float* part, float* bpart, int S, hipStream_t st) {
  const int nz = conv2_wgrad_frames_splits(S);
  const int fpb = (S + nz - 1) / nz;
  hipLaunchKernelGGL(conv2_wgrad_frames_kernel, dim3(nz), dim3(256), 0, st, act1, dypad, part, bpart, S, fpb);
}

// ------------------------------------------------------------------------------------------ conv3: act2 [9][9][64] x dY3 [7][7][64]
// x-tiles: 9 taps x 2 ci halves, y-tiles: 2 co halves = 36 accumulator tiles; wave (c, j) = (wave >> 1, wave & 1) owns the 9 taps of ci
// half c for co half j.  LDS: two stages of {act2 20,736 B + dY interior 12,544 B} = 66,560 B -> two blocks per CU, frame s+1 in flight
// while frame s is multiplied.
#define C3W_A_FLOATS 5184
#define C3W_B_FLOATS 3136
#define C3W_STAGE (C3W_A_FLOATS + C3W_B_FLOATS)
__global__ __launch_bounds__(256, 2) void conv3_wgrad_frames_kernel(const float* act2, const float* dypad, float* part, float* bpart, int S,
                                                                    int frames_per_block) {
  __shared__ __attribute__((aligned(16))) float smem[2 * C3W_STAGE];   // stage: [ih*9 + iw][ci] | [oh*7 + ow][co]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = wave >> 1, j = wave & 1;
  const int s_lo = blockIdx.x * frames_per_block, s_hi = min(S, s_lo + frames_per_block);
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
  float bs = 0.0f;
  auto stage = [&](int buf, int s) __attribute__((always_inline)) {
    float* As = smem + buf * C3W_STAGE;
    float* Bs = As + C3W_A_FLOATS;
    slab_to_lds<C3W_A_FLOATS / 4>(act2 + (size_t)s * C3W_A_FLOATS, As, wave, lane);
    const float* src = dypad + (size_t)s * 7744;   // 7 rows of 7*64 floats (112 pieces each) at border 2
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {   // 784 pieces
      const int v0 = (wave + 4 * r) * 64, v = v0 + lane;
      if (v < 784) {
        const int row = v / 112, col = v - row * 112;
        glds16(reinterpret_cast<const char*>(src) + (unsigned)((((2 + row) * 11 + 2) * 64) * 4 + col * 16), reinterpret_cast<char*>(Bs) + (unsigned)v0 * 16u);
      }
    }
  };
  if (s_lo < s_hi) stage(0, s_lo);
  for (int s = s_lo; s < s_hi; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < s_hi) stage((s + 1 - s_lo) & 1, s + 1);
    const float* As = smem + ((s - s_lo) & 1) * C3W_STAGE;
    const float* Bs = As + C3W_A_FLOATS;
    const float* an = As + 32 * c + li + h * 64;              // next ow: +1 pixel * 64 channels
    const float* aw = As + 32 * c + li + h * (3 * 64);        // ow = 6 -> (oh + 1, 0): +9 - 6 pixels
    const float* al = As + 32 * c + li;                       // step 24: both halves read position 48 (h = 1 gets B = 0)
    const float* bl = Bs + 32 * j + li + h * 64;
    // 49 positions = 24 pairs + 1 single
    float fa[2][9], fb[2];
    auto frag = [&](int q, int set) __attribute__((always_inline)) {
      const int p0 = 2 * q, oh = p0 / 7, ow = p0 - oh * 7;
      const float* ap = (q == 24 ? al : (ow == 6 ? aw : an)) + (oh * 9 + ow) * 64;
#pragma unroll
      for (int t = 0; t < 9; ++t) fa[set][t] = ap[((t / 3) * 9 + (t % 3)) * 64];
      if (q < 24) fb[set] = bl[p0 * 64];
      else { const float v = Bs[48 * 64 + 32 * j + li]; fb[set] = h ? 0.0f : v; }
    };
    frag(0, 0);
#pragma unroll
    for (int q = 0; q < 25; ++q) {
      if (q + 1 < 25) frag(q + 1, (q + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const float b = fb[q & 1];
      if (c == 0) bs += b;
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][t], b, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // partial out: part[block][(kh*3 + kw)*64 + 32*c + row][32*j + li]
  float* o = part + (size_t)blockIdx.x * 576 * 64;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      o[(size_t)(t * 64 + 32 * c + row) * 64 + 32 * j + li] = acc[t][e];
    }
  if (c == 0) {
    bs += __shfl_xor(bs, 32, 64);
    if (h == 0) bpart[(size_t)blockIdx.x * 64 + 32 * j + li] = bs;
  }
}

static const int C3W_BLOCKS = 256;   // one block per CU (two fit): 146 -> 127 us and half the partials of 512
int conv3_wgrad_frames_splits(int S) {
  int blocks = C3W_BLOCKS;
  if (S < blocks) blocks = S;
  const int fpb = (S + blocks - 1) / blocks;
  return (S + fpb - 1) / fpb;
}
int conv3_wgrad_frames_splits_bound(int maxS) { return maxS < C3W_BLOCKS ? maxS : C3W_BLOCKS; }
void launch_conv3_wgrad_frames(const float* act2, const float* dypad, float* part, float* bpart, int S, hipStream_t st) {
  const int nz = conv3_wgrad_frames_splits(S);
  const int fpb = (S + nz - 1) / nz;
  hipLaunchKernelGGL(conv3_wgrad_frames_kernel, dim3(nz), dim3(256), 0, st, act2, dypad, part, bpart, S, fpb);
}
