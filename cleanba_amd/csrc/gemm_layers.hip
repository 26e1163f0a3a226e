// gemm_layers.hip — Nature-CNN forward / backward as implicit GEMMs on the f32 MFMA
// (problem functors for igemm.h + the layer drivers).
//
// Reference: Network naturecnn:143-178, Actor/Critic ppo:192-203; backward = what
// jax.value_and_grad emits at ppo:590,619 / impala:607.
//
// HBM layout: activations NHWC fp32 (= flax's layout, so conv3's output
// is already the (h,w,c) flatten of ppo:185); frames stay uint8 NCHW exactly as the env wrote
// them (the transpose + /255 of ppo:180-181 is folded into conv1's tile load); pre-activation
// gradients of conv2/conv3 outputs live in zero-bordered 11x11 buffers so that both dgrads are
// plain VALID correlations (stride-2 conv2 as 4 parity classes).
#include "cbm_internal.h"
// (Tile order: igemm.h ORDER 2 — the four parity classes of a conv2 dgrad pixel tile back to back on one XCD — is what the merged position-major
// conv2 dgrad does by construction; ORDER 1 for the weight-gradient tap tiles measured slower and is gone with the im2col wgrads.)
#ifdef CBM_BLOCK_TRACE   // timing build: wall-clock stamps of every block of the last igemm_dma_kernel launch (the dense forward), tools/block_trace.py
__device__ unsigned long long cbm_block_trace_dma[512][2];
extern "C" int cbm_debug_block_trace_dma(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(cbm_block_trace_dma), sizeof(cbm_block_trace_dma)) == hipSuccess ? 0 : -1; }
#define IG_BT_START() do { if (threadIdx.x == 0 && blockIdx.x < 512) cbm_block_trace_dma[blockIdx.x][0] = wall_clock64(); } while (0)
#define IG_BT_END() do { if (threadIdx.x == 0 && blockIdx.x < 512) cbm_block_trace_dma[blockIdx.x][1] = wall_clock64(); } while (0)
#endif
#include "igemm.h"
#include "env_model.h"
#include "ppo_loss.h"
#include <algorithm>
#include <vector>
#include <type_traits>
#include <typeinfo>
#include <string>
#include <cxxabi.h>

NatureLayout nature_layout(int A) {
  NatureLayout L;
  L.A = A;
  const int64_t wsz[6] = {8 * 8 * 4 * 32, 4 * 4 * 32 * 64, 3 * 3 * 64 * 64, 3136 * 512, 512 * (int64_t)A, 512};
  const int64_t bsz[6] = {32, 64, 64, 512, A, 1};
  int64_t o = 0;
  for (int i = 0; i < 6; ++i) { L.w[i] = o; o += wsz[i]; L.b[i] = o; o += bsz[i]; }
  L.total = o;
  return L;
}

static __device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
static __device__ __forceinline__ float relu(float v) { return v > 0.0f ? v : 0.0f; }
// Loads are always issued (clamped address) and masked afterwards: a branch around a load makes hipcc
// wait vmcnt(0) at every merge point and serialises the gather into dependent round trips.
static __device__ __forceinline__ float4 f4sel(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// ReLU masks as bits.  In every igemm epilogue call the 32 lanes of a wave half hold 32 consecutive columns of ONE output row, so a
// ballot gives that row's mask word; lane 0 of the half writes it.  words_per_row = channels / 32.
static __device__ __forceinline__ void put_mask_word(uint32_t* mask, size_t row, int words_per_row, int col, bool on) {
  const unsigned long long bal = __ballot(on);
  const int lane = threadIdx.x & 63;
  if ((lane & 31) == 0) mask[row * words_per_row + (col >> 5)] = (uint32_t)(bal >> (lane & 32));
}

// ------------------------------------------------------------------------------------------
// conv1: uint8 NCHW frames -> act1 [M=S*400][32].  k = (c, kh, kw), 8 contiguous bytes per (c,kh).
template <class TileT>
struct Conv1Fwd {
  using Tile = TileT;
  static constexpr bool A_RX = false, B_YR = false, BIAS_GRAD = false;
  static constexpr int NCLS = 1;
  const uint8_t* obs; const int32_t* idx; const float* W; const float* bias; float* out; int M; uint32_t* mask;
  __host__ __device__ int X() const { return M; }
  __host__ __device__ int Y() const { return 32; }
  __device__ void r_range(int, int& lo, int& hi) const { lo = 0; hi = 256; }
  __device__ float4 load_a(int m, int r, int, int) const {
    m = min(m, M - 1);  // rows >= M are never stored
    const int s = m / 400, p = m - s * 400, oh = p / 20, ow = p - oh * 20;
    const int f = idx ? idx[s] : s;
    const int c = r >> 6, kh = (r >> 3) & 7, kw = r & 7;
    const uint32_t w = *reinterpret_cast<const uint32_t*>(obs + (size_t)f * CBM_FRAME + c * 7056 + (oh * 4 + kh) * 84 + ow * 4 + kw);
    return make_float4(cbm_u8_unit(w & 255u), cbm_u8_unit((w >> 8) & 255u), cbm_u8_unit((w >> 16) & 255u), cbm_u8_unit(w >> 24));
  }
  __device__ float4 load_b(int r, int y, int, int) const {
    const int c = r >> 6, kh = (r >> 3) & 7, kw = r & 7;
    return *reinterpret_cast<const float4*>(W + ((kh * 8 + kw) * 4 + c) * 32 + y);
  }
  // row / chunk split of the same addresses for the small-batch kernel (K chunks of 32 = 4 patch rows of 8 pixels, 64 = one channel plane's
  // patch): A offsets are BYTES into obs, B offsets floats into the HWIO weights
  static constexpr bool ROWPTR_S16 = true;
  __device__ uint32_t a_off(int m, int rl, int) const {
    m = min(m, M - 1);
    const int s = m / 400, p = m - s * 400, oh = p / 20, ow = p - oh * 20;
    const int f = idx ? idx[s] : s;
    return (uint32_t)(f * CBM_FRAME + (oh * 4 + (rl >> 3)) * 84 + ow * 4 + (rl & 7));
  }
  __device__ uint32_t a_chunk(int r0) const { return (uint32_t)((r0 >> 6) * 7056 + ((r0 >> 3) & 7) * 84); }
  __device__ uint32_t b_off(int rl, int y, int) const { return (uint32_t)((((rl >> 3) * 8 + (rl & 7)) * 4) * 32 + y); }
  __device__ uint32_t b_chunk(int r0) const { return (uint32_t)(((((r0 >> 3) & 7) * 8) * 4 + (r0 >> 6)) * 32); }
  __device__ float4 rp_a(uint32_t off) const {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(obs + off);
    return make_float4(cbm_u8_unit(w & 255u), cbm_u8_unit((w >> 8) & 255u), cbm_u8_unit((w >> 16) & 255u), cbm_u8_unit(w >> 24));
  }
  __device__ float4 rp_b(uint32_t off) const { return ig_ld4(W + off); }
  __device__ void store(int m, int n, float v, int, int) const {
    if (m < M) {
      const float o = relu(v + bias[n]);
      out[(size_t)m * 32 + n] = o;
      if (mask) put_mask_word(mask, (size_t)m, 1, n, o > 0.0f);
    }
  }
  static constexpr bool BIAS_PRE = true;   // (small-batch kernel: no masks there)
  __device__ float bias_pre(int n) const { return bias[n]; }
  __device__ void store_pre(int m, int n, float v, float b) const { if (m < M) out[(size_t)m * 32 + n] = relu(v + b); }
  static constexpr bool MASKOUT = true;   // igemm_kernel epilogue: one mask store per 32x32 tile instead of one per element row
  __device__ bool store_flag(int m, int n, float v, int, int) const {
    const float o = relu(v + bias[n]);
    if (m < M) out[(size_t)m * 32 + n] = o;
    return o > 0.0f;
  }
  __device__ void put_mask(int m, int, uint32_t word) const { if (mask && m < M) mask[m] = word; }
  // write-through (sc1) form of store_pre for the dataflow actor step: the consumer is another workgroup of the SAME launch (actor_fused_kernel)
  __device__ void store_wt(int m, int n, float v, float b, int) const { if (m < M) cbm_store_wt(out + (size_t)m * 32 + n, relu(v + b)); }
  __device__ const void* a_origin_wt() const { return obs; }   // (frames come from the previous launch: plain loads)
};

// generic VALID NHWC conv forward, k = (kh, kw, ci)
template <class TileT, int KH, int KW, int ST, int CI, int CO, int IH, int IW, int OH, int OW>
struct ConvFwd {
  using Tile = TileT;
  static constexpr bool A_RX = false, B_YR = false, BIAS_GRAD = false;
  static constexpr int NCLS = 1;
  const float* in; const float* W; const float* bias; float* out; int M; uint32_t* mask;
  __host__ __device__ int X() const { return M; }
  __host__ __device__ int Y() const { return CO; }
  __device__ void r_range(int, int& lo, int& hi) const { lo = 0; hi = KH * KW * CI; }
  __device__ float4 load_a(int m, int r, int, int) const {
    m = min(m, M - 1);
    const int s = m / (OH * OW), p = m - s * (OH * OW), oh = p / OW, ow = p - oh * OW;
    const int kh = r / (KW * CI), rem = r - kh * (KW * CI);
    return *reinterpret_cast<const float4*>(in + ((size_t)(s * IH + oh * ST + kh) * IW + ow * ST) * CI + rem);
  }
  __device__ float4 load_b(int r, int y, int, int) const { return *reinterpret_cast<const float4*>(W + (size_t)r * CO + y); }
  // row / chunk split of the same addresses (igemm.h ROWPTR): a chunk never leaves one kernel row (KW*CI is a multiple of the K chunk)
  // (used by the small-batch kernel only: on igemm_pf2_kernel the split measured neutral for the conv3 forward and slower for the conv2 forward)
  static constexpr bool ROWPTR_S16 = true;   // K chunks of 32 / 64 never leave a kernel row
  __device__ float4 rp_a(uint32_t off) const { return ig_ld4(in + off); }
  __device__ float4 rp_b(uint32_t off) const { return ig_ld4(W + off); }
  __device__ const float* a_origin() const { return in; }
  __device__ const float* b_origin() const { return W; }
  __device__ uint32_t a_off(int m, int rl, int) const {
    m = min(m, M - 1);
    const int s = m / (OH * OW), p = m - s * (OH * OW), oh = p / OW, ow = p - oh * OW;
    return (uint32_t)(((s * IH + oh * ST) * IW + ow * ST) * CI + rl);
  }
  __device__ uint32_t a_chunk(int r0) const { const int kh = r0 / (KW * CI); return (uint32_t)(kh * IW * CI + (r0 - kh * (KW * CI))); }
  __device__ uint32_t b_off(int rl, int y, int) const { return (uint32_t)(rl * CO + y); }
  __device__ uint32_t b_chunk(int r0) const { return (uint32_t)(r0 * CO); }
  static constexpr bool DMA_OK = false;   // (igemm_dma_kernel measured slower for the convs: conv3 148 -> 194 us; faster for the dense layer)
  __device__ void store(int m, int n, float v, int, int) const {
    if (m < M) {
      const float o = relu(v + bias[n]);
      out[(size_t)m * CO + n] = o;
      if (mask) put_mask_word(mask, (size_t)m, CO / 32, n, o > 0.0f);
    }
  }
  static constexpr bool BIAS_PRE = true;   // (small-batch kernel: no masks there)
  __device__ float bias_pre(int n) const { return bias[n]; }
  __device__ void store_pre(int m, int n, float v, float b) const { if (m < M) out[(size_t)m * CO + n] = relu(v + b); }
  static constexpr bool MASKOUT = true;
  __device__ bool store_flag(int m, int n, float v, int, int) const {
    const float o = relu(v + bias[n]);
    if (m < M) out[(size_t)m * CO + n] = o;
    return o > 0.0f;
  }
  __device__ void put_mask(int m, int n32, uint32_t word) const { if (mask && m < M) mask[(size_t)m * (CO / 32) + (n32 >> 5)] = word; }
  __device__ void store_wt(int m, int n, float v, float b, int) const { if (m < M) cbm_store_wt(out + (size_t)m * CO + n, relu(v + b)); }   // (see Conv1Fwd)
  __device__ const void* a_origin_wt() const { return in; }
};

// dense forward C[m][n] = sum_k A[m][k] W[k][n]; SPLIT: partials [z][M][N], else relu(+bias)
static __device__ __forceinline__ float4 f4relu(float4 v) { return make_float4(relu(v.x), relu(v.y), relu(v.z), relu(v.w)); }
template <class TileT, bool SPLIT, bool PRE_RELU = false>
struct DenseFwd {
  using Tile = TileT;
  static constexpr bool A_RX = false, B_YR = false, BIAS_GRAD = false;
  static constexpr int NCLS = 1;
  const float* A; const float* W; const float* bias; float* out; int M, K, N, seg;
  __host__ __device__ int X() const { return M; }
  __host__ __device__ int Y() const { return N; }
  __device__ void r_range(int z, int& lo, int& hi) const { lo = z * seg; hi = lo + seg; }
  __device__ float4 load_a(int m, int r, int rhi, int) const {
    const bool ok = r < rhi;
    const float4 v = f4sel(ok, *reinterpret_cast<const float4*>(A + (size_t)min(m, M - 1) * K + min(r, K - 4)));
    return PRE_RELU ? f4relu(v) : v;
  }
  __device__ float4 load_b(int r, int y, int rhi, int) const {
    const bool ok = r < rhi && y < N;
    return f4sel(ok, *reinterpret_cast<const float4*>(W + (size_t)min(r, K - 1) * N + min(y, N - 4)));
  }
  static constexpr bool DMA_OK = !PRE_RELU;   // needs seg % BR == 0 and N % BY == 0 (checked at the call site)
  static constexpr bool ROWPTR_S16 = !PRE_RELU;   // same preconditions (whole chunks, whole column tiles)
  __device__ float4 rp_a(uint32_t off) const { return ig_ld4(A + off); }
  __device__ float4 rp_b(uint32_t off) const { return ig_ld4(W + off); }
  __device__ const float* a_origin() const { return A; }
  __device__ const float* b_origin() const { return W; }
  __device__ uint32_t a_off(int m, int rl, int) const { return (uint32_t)(min(m, M - 1) * K + rl); }
  __device__ uint32_t a_chunk(int r0) const { return (uint32_t)r0; }
  __device__ uint32_t b_off(int rl, int y, int) const { return (uint32_t)(rl * N + y); }
  __device__ uint32_t b_chunk(int r0) const { return (uint32_t)(r0 * N); }
  __device__ const float* a_ptr(int m, int r, int) const { return A + (size_t)min(m, M - 1) * K + r; }
  __device__ const float* b_ptr(int r, int y, int) const { return W + (size_t)r * N + y; }
  __device__ void store(int m, int n, float v, int z, int) const {
    if (m >= M || n >= N) return;
    if (SPLIT) out[((size_t)z * M + m) * N + n] = v;
    else out[(size_t)m * N + n] = relu(v + bias[n]);
  }
  static constexpr bool BIAS_PRE = !SPLIT;
  __device__ float bias_pre(int n) const { return bias[min(n, N - 1)]; }
  __device__ void store_pre(int m, int n, float v, float b) const { if (m < M && n < N) out[(size_t)m * N + n] = relu(v + b); }
  __device__ void store_wt(int m, int n, float v, float, int z) const {   // split-K partial, write-through (the next LAUNCH reads it: plain stores would do; kept uniform)
    if (m < M && n < N) out[((size_t)z * M + m) * N + n] = v;
  }
  __device__ const void* a_origin_wt() const { return A; }
};

__global__ void dense_reduce_kernel(const float* part, const float* bias, float* out, int M, int N, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  float t = part[i];
  for (int s = 1; s < S; ++s) t = t + part[(size_t)s * M * N + i];
  out[i] = relu(t + bias[i % N]);
}

// heads: logits[b][a] = chain_k hid[b][k] Wa[k][a] + ba[a]; value likewise.  One thread per output, the
// 512-long fmaf chain stays serial (numerics spec); hid rows and both weight matrices are staged in LDS so the
// chain is paced by the FMA latency, not by global loads.
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* hid, const float* Wa, const float* ba, const float* Wc,
                                                        const float* bc, int B, int A, int HD, float* logits, float* value) {
  extern __shared__ __attribute__((aligned(16))) float hsm[];
  float* hs = hsm;             // [8][HD]
  float* ws = hsm + 8 * HD;    // [HD][A+1]
  const int A1 = A + 1;
  const int f0 = blockIdx.x * 8;
  for (int i = threadIdx.x; i < 8 * HD; i += 256) {
    const int f = f0 + i / HD;
    hs[i] = f < B ? hid[(size_t)f * HD + i % HD] : 0.0f;
  }
  for (int i = threadIdx.x; i < HD * A; i += 256) ws[(i / A) * A1 + i % A] = Wa[i];
  for (int i = threadIdx.x; i < HD; i += 256) ws[i * A1 + A] = Wc[i];
  __syncthreads();
  const int fl = threadIdx.x >> 5, o = threadIdx.x & 31, f = f0 + fl;
  if (f >= B || o > A) return;
  const float* h = hs + fl * HD;
  const float* w = ws + o;
  float acc = 0.0f;
#pragma unroll 8
  for (int k = 0; k < HD; ++k) acc = fmaf(h[k], w[k * A1], acc);
  if (o < A) logits[(size_t)f * A + o] = acc + ba[o];
  else value[f] = acc + bc[0];
}

// The same heads as ONE small GEMM on the 16x16x4 kernel (igemm.h igemm_s16_kernel): C[m][n] = chain_k hid[m][k] * [Wa | Wc][k][n], n < A the
// logits, n == A the value; the instruction's k-ascending accumulation is the serial fmaf chain above, bias added afterwards -> same bits,
// 512 dependent FMAs (and 55 KB of LDS staging per 8 rows) become 128 MFMAs of 32 cycles per 16x16 tile: 16.4 -> ~6 us per call.
struct HeadsFwd {
  using Tile = IgemmTile<64, 64, 16, 2, 2>;   // (unused by igemm_s16_kernel)
  static constexpr bool A_RX = false, B_YR = false, BIAS_GRAD = false;
  static constexpr int NCLS = 1;
  const float* hid; const float* Wa; const float* ba; const float* Wc; const float* bc; float* logits; float* value; int M, A, HD;
  __host__ __device__ int X() const { return M; }
  __host__ __device__ int Y() const { return A + 1; }
  __device__ void r_range(int, int& lo, int& hi) const { lo = 0; hi = HD; }
  __device__ float4 load_a(int m, int r, int, int) const { return *reinterpret_cast<const float4*>(hid + (size_t)min(m, M - 1) * HD + r); }
  __device__ float wcol(int r, int n) const { return n < A ? Wa[r * A + n] : (n == A ? Wc[r] : 0.0f); }
  __device__ float4 load_b(int r, int y, int, int) const { return make_float4(wcol(r, y), wcol(r, y + 1), wcol(r, y + 2), wcol(r, y + 3)); }
  __device__ void store(int m, int n, float v, int, int) const {
    if (m >= M) return;
    if (n < A) logits[(size_t)m * A + n] = v + ba[n];
    else if (n == A) value[m] = v + bc[0];
  }
};
// Actor tail: heads + sampling in ONE launch.  One block per 16 frames: their hid rows go to LDS, the head weights to registers, the heads run on v_mfma_f32_16x16x4_f32 (HeadsFwd's chain: k ascending from 0, bias afterwards), logits / value stay
// in LDS, and the sampling is sample_kernel's code (pointwise.hip: same threefry counters, same shuffle tournament, same exp-sum order).
// Replaces the heads GEMM + sample_kernel (12.9 + 4.8 us per actor step); logits / value never reach HBM.  Bit-identical outputs.
template <int HD>
__global__ __launch_bounds__(256) void actor_tail_kernel(const float* hid, const float* Wa, const float* ba, const float* Wc, const float* bc, int B, int A,
                                                         ActorSample smp) {
  constexpr int PH = HD + 4;                       // row pitch: conflict-free A fragment reads (lane (g4, r16): hs[r16][4s + g4])
  __shared__ __attribute__((aligned(16))) float hs[16 * PH];     // hid rows
  __shared__ float lg[16][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g4 = lane >> 4;
  const int m0 = blockIdx.x * 16;
  // waves 0 / 1 own output columns 0-15 / 16-31: ALL their B fragments (HD/4 floats per lane, L2-resident weights) are requested up front,
  // together with the hid rows, so the block pays one load latency instead of one per K chunk
  const int n = wave * 16 + r16;
  const bool on = wave < 2 && n <= A;
  const float* wp = n < A ? Wa + n : Wc;
  const int wstride = n < A ? A : 1;
  float bw[HD / 4];
#pragma unroll
  for (int st = 0; st < HD / 4; ++st) bw[st] = on ? wp[(size_t)(4 * st + g4) * wstride] : 0.0f;
  for (int i = tid; i < 16 * HD / 4; i += 256) {
    const int row = i / (HD / 4), c4 = (i % (HD / 4)) * 4;
    *reinterpret_cast<float4*>(hs + row * PH + c4) = *reinterpret_cast<const float4*>(hid + (size_t)min(m0 + row, B - 1) * HD + c4);
  }
  __syncthreads();
  if (wave < 2) {
    f32x4_mfma acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int st = 0; st < HD / 4; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hs[r16 * PH + 4 * st + g4], bw[st], acc, 0, 0, 0);
    const float bias = n < A ? ba[n] : (n == A ? bc[0] : 0.0f);
#pragma unroll
    for (int e = 0; e < 4; ++e) lg[4 * g4 + e][n] = acc[e] + bias;   // D layout: lane (g4, r16) holds rows 4*g4 + e, column r16
  }
  __syncthreads();
  const uint32_t nn = (uint32_t)(B * A);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {   // sampling: 32 lanes per row, 8 rows per pass
    const int row = pass * 8 + (tid >> 5), b = m0 + row, a = tid & 31;
    const bool live = b < B && a < A;
    const int bb = b < B ? b : B - 1, aa = a < A ? a : A - 1;
    const float z = lg[row][aa];
    const float u = cbm_bits_to_uniform(cbm_random_bits_at(smp.sk0, smp.sk1, nn, (uint32_t)(bb * A + aa)));
    float g = live ? z - cbm_logf(-cbm_logf(u)) : -INFINITY;
    if (live && smp.logits_out) smp.logits_out[(size_t)b * A + a] = z;
    int bi = a;
    float bv = g, mx = live ? z : -INFINITY;
    for (int o = 16; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 32);
      const int oi = __shfl_xor(bi, o, 32);
      const float om = __shfl_xor(mx, o, 32);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      mx = om > mx ? om : mx;
    }
    const float e = live ? cbm_expf(z - mx) : 0.0f;
    const float zb = __shfl(z, bi, 32);
    float ssum = 0.0f;
    for (int j = 0; j < A; ++j) ssum += __shfl(e, j, 32);
    if (b < B && a == 0) {
      smp.actions[b] = bi;
      if (smp.logprobs) smp.logprobs[b] = (zb - mx) - cbm_logf(ssum);
      if (smp.value_out) smp.value_out[b] = lg[row][A];
    }
  }
}
// Actor tail, one block per frame: the dense layer's split-K reduction, the two heads and the sampling in ONE launch.
// The 16-row MFMA tail above needs the reduced hidden rows in HBM first (dense_reduce_kernel, 4.9 us + a launch); folding the reduction into
// its 8 blocks made them the bottleneck (15.8 ms rollouts).  With a block per frame the reduction is 120-way parallel again: every thread
// sums its two hidden units over the S partial slices in slice order (dense_reduce_kernel's chain), the A + 1 head outputs are 512-long
// k-ascending fmaf chains on A + 1 lanes (bitwise the MFMA's chain: DESIGN 3) fed from LDS, and the frame's 32 sampling lanes run
// sample_kernel's code.  hid never reaches HBM.
#ifdef CBM_S16_TRACE
extern "C" int cbm_debug_s16_trace(int sel_x, unsigned long long* out) {
  if (sel_x >= 0) return hipMemcpyToSymbol(HIP_SYMBOL(cbm_s16_trace_sel), &sel_x, sizeof(int)) == hipSuccess ? 0 : -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cbm_s16_trace), sizeof(cbm_s16_trace)) == hipSuccess ? 0 : -1;
}
#endif
// timing build only (tools/variants.sh tailtrace "-DCBM_TAIL_TRACE"): shader-clock stamps of block 0's four waves at the phase boundaries of the tail
#ifdef CBM_TAIL_TRACE
__device__ unsigned long long cbm_tail_trace[4][16];
extern "C" int cbm_debug_tail_trace(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(cbm_tail_trace), sizeof(cbm_tail_trace)) == hipSuccess ? 0 : -1; }
#define TT(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cbm_tail_trace[threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TT(k) do { } while (0)
#endif
template <int HD>
__global__ __launch_bounds__(256) void actor_tail_rows_kernel(const float* part, const float* bd, int S, const float* Wa, const float* ba, const float* Wc,
                                                              const float* bc, int B, int A, ActorSample smp) {
  __builtin_amdgcn_s_setprio(3);   // (see igemm_s16_kernel)
  TT(0);
  __shared__ int act_s;
  __shared__ EnvShared env_sh;
  __shared__ __attribute__((aligned(16))) float hsT[HD];      // hid of this frame, stored as [k % 4][k / 4]: lane group g4 of a 16x16x4 MFMA reads its k = 4*st + g4 as consecutive floats
  __shared__ __attribute__((aligned(16))) float wl[HD * 28];  // actor weights [HD][A] as they lie in memory (A <= 28)
  __shared__ float wcl[HD];                                   // critic weights
  __shared__ float lg[32], gum[32];
  const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, r16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t MN = (size_t)B * HD;
  // device env: this block also steps env b with the action it samples.  Almost all of that step does not need the action (env_model.h) and runs on
  // the waves the heads leave idle: the planes that survive the shift are requested now by waves 1-3 (16 bytes per load); wave 2 computes the three
  // transitions an action can cause while waves 0 / 1 run the heads' MFMA chain; waves 1-3 store the shifted stack and the new plane (= the previous
  // newest plane with the ball moved) while wave 0 samples; what is left after the action is the choice among the three transitions and the eleven
  // pieces that hold the paddle rows
  const EnvStepArgs ea{smp.env_seed, smp.env_max_steps, smp.env_st, smp.env_obs_prev, smp.env_obs_next, smp.env_reward, smp.env_done_next,
                       smp.env_firststep_next};
  const bool env_cand = ea.obs_next && tid >= 128 && tid < 131;
  uint32_t env_s0[ENV_STATE_WORDS];
  if (env_cand) env_state_load_words(ea.st, b, env_s0);
  EnvPieces<192> env_pc;
  if (ea.obs_next && wave >= 1) env_step_prefetch<192>(ea, b, tid - 64, env_pc);
  TT(11);
  // Division of the load phase: waves 0 / 1 bring the head weights into LDS with the load unit (global_load_lds: 1 KB per wave instruction, no
  // registers: [HD][A] is HD / 256 * A instructions of 64 x 16 bytes, the critic column HD / 64 of 64 x 4 bytes) and, after a first barrier, lift
  // their B fragments out of LDS while waves 2 / 3 still wait for the dense layer's partial slices (all S <= 16 slices of a thread's HD / 128 hidden
  // units are requested at once, then added in slice order).  (Held in registers straight from L2 — 128 strided loads per lane — the fragments' ISSUE
  // alone took 3.2 us of a 5 us load phase: in-kernel clock stamps, tools/tail_trace.py.)
  constexpr int NU = HD / 128;
  static_assert(HD % 256 == 0 && NU >= 2 && NU <= 4, "hidden width 256 or 512");
  float vv[NU][16];
  if (wave >= 2) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int ss = s < S ? s : 0;
#pragma unroll
      for (int u = 0; u < NU; ++u) vv[u][s] = part[ss * MN + (size_t)b * HD + (tid - 128) + 128 * u];
    }
  } else {
    if (((uintptr_t)Wa & 15) == 0) {
      for (int q = wave; q < (HD / 256) * A; q += 2) ig_glds16(Wa + (size_t)(64 * q + lane) * 4, wl + 64 * q * 4);
    } else {
      for (int i = tid; i < HD * A; i += 128) wl[i] = Wa[i];
    }
    for (int q = wave; q < HD / 64; q += 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wc + 64 * q + lane), (__attribute__((address_space(3))) void*)(wcl + 64 * q), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the weights has landed in LDS
  }
  TT(12);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (raw: waves 2 / 3 arrive with their loads in flight)
  TT(1);
  const int hn = wave * 16 + r16;                    // waves 0 / 1: this lane's head column
  float bw[HD / 4];
  if (wave < 2) {
    // columns past A + 1 multiply whatever the critic column holds: their results are never read, and no lane needs a predicate around its reads;
    // a running pointer, because wq[st * wstep] is a quarter-rate v_mul_lo_u32 per read (2 us for the 128 of them)
    const float* wr = (hn < A ? wl + hn : wcl) + g4 * (hn < A ? A : 1);
    const int wstep = 4 * (hn < A ? A : 1);
#pragma unroll
    for (int st = 0; st < HD / 4; ++st) { bw[st] = *wr; wr += wstep; }
    TT(13);
  } else {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      float t0 = vv[u][0];
#pragma unroll
      for (int s = 1; s < 16; ++s)
        if (s < S) t0 = t0 + vv[u][s];
      const int k = (tid - 128) + 128 * u;
      hsT[(k & 3) * (HD / 4) + (k >> 2)] = relu(t0 + bd[k]);
    }
  }
  TT(3);
  __syncthreads();
  TT(4);
  if (wave < 2) {
    // heads on v_mfma_f32_16x16x4_f32 with ONE live row (row 0 = this frame; the instruction's other 15 rows repeat it and are ignored): waves
    // 0 / 1 own output columns 0-15 / 16-31; column n <= A is a 512-long k-ascending chain (bitwise the 16-row tail's and heads_fwd's: DESIGN 3).
    // Every fragment is in registers before the first MFMA (read inside the chain, each group of MFMAs waited for its own LDS round trip: 5.0 us against 2.4)
    const int n = hn;
    float4 hq[HD / 16];
#pragma unroll
    for (int q = 0; q < HD / 16; ++q) hq[q] = *reinterpret_cast<const float4*>(hsT + g4 * (HD / 4) + 4 * q);   // k = 4*(4q + i) + g4, i = 0..3
    __builtin_amdgcn_sched_barrier(0);     // (every fragment is requested before the first MFMA: inside the chain each group of eight waited for its own LDS round trip)
    TT(14);
    f32x4_mfma acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < HD / 16; ++q) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hq[q].x, bw[4 * q], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hq[q].y, bw[4 * q + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hq[q].z, bw[4 * q + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hq[q].w, bw[4 * q + 3], acc, 0, 0, 0);
    }
    if (g4 == 0 && n <= A) lg[n] = acc[0] + (n < A ? ba[n] : bc[0]);   // D: lane (g4 = 0, r16) element 0 = row 0, column r16
    TT(6);
  } else if (wave == 2) {
    if (env_cand) env_step_candidates(ea, b, env_sh, tid - 128, env_s0);   // the three transitions an action can cause
  } else if (lane < 32) {
    // the Gumbel perturbation of every action: it depends on the key and the frame, not on the logits
    const int aa = lane < A ? lane : A - 1;
    const float u = cbm_bits_to_uniform(cbm_random_bits_at(smp.sk0, smp.sk1, (uint32_t)(B * A), (uint32_t)(b * A + aa)));
    gum[lane] = cbm_logf(-cbm_logf(u));
  }
  TT(5);
  __syncthreads();
  TT(7);
  if (tid < 32) {
    const int a = tid;
    const bool live = a < A;
    const int aa = live ? a : A - 1;
    const float z = lg[aa];
    float g = live ? z - gum[a] : -INFINITY;      // z - log(-log(u)): jax.random.categorical's Gumbel arg-max (ppo:256-259)
    if (live && smp.logits_out) smp.logits_out[(size_t)b * A + a] = z;
    int bi = a;
    float bv = g, mx = live ? z : -INFINITY;
    for (int o = 16; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 32);
      const int oi = __shfl_xor(bi, o, 32);
      const float om = __shfl_xor(mx, o, 32);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      mx = om > mx ? om : mx;
    }
    const float e = live ? cbm_expf(z - mx) : 0.0f;
    const float zb = __shfl(z, bi, 32);
    float ej[28];
#pragma unroll
    for (int j = 0; j < 28; ++j) ej[j] = __shfl(e, j, 32);      // (all requested before the first add: one LDS-crossbar latency instead of A)
    float ssum = 0.0f;
#pragma unroll
    for (int j = 0; j < 28; ++j)
      if (j < A) ssum += ej[j];
    if (a == 0) {
      smp.actions[b] = bi;
      if (ea.obs_next) act_s = env_action_dir(env_sh, bi);      // what the env's finish needs of the action: the paddle direction
      if (smp.logprobs) smp.logprobs[b] = (zb - mx) - cbm_logf(ssum);
      if (smp.value_out) smp.value_out[b] = lg[A];
    }
  } else if (ea.obs_next && wave >= 1) {
    env_step_early<192>(ea, b, env_sh, tid - 64, env_pc);     // under wave 0's sampling
  }
  TT(8);
  if (ea.obs_next) {
    // (a raw barrier: the action travels through LDS, and the early part's stores need not have landed — a piece the finish stores again is stored
    // by the thread that stored it before; __syncthreads() also waits for vmcnt(0), 1.2 us of store latency on the block's critical path)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TT(9);
    if (wave >= 1) env_step_finish<192>(ea, b, env_sh, act_s, tid - 64, env_pc);
  }
  TT(10);
}
// ================================================================================================ dataflow actor step (round 6)
// An actor step was five dependent launches of ~10 us for 2.2 GFLOP (conv1, conv2, conv3, split-K dense, per-frame tail): every boundary drains the chip,
// costs 1.5-1.9 us of its own and makes the next layer's blocks start cold.  Here conv1, conv2, conv3 and the dense layer are ONE launch: the grid is the
// concatenation of the four layers' block lists in layer order, every block runs igemm_s16_kernel's body (same tiles, same k-ascending 16x16x4 MFMA chain ->
// the same bits), and instead of a kernel boundary a consumer block waits for the FRAMES its rows read: per layer and frame a monotonic arrival counter
// counts the (row, column-block) pieces stored so far; launch number `epoch` is complete for a frame at epoch * pieces-per-frame (never reset, wrap-safe
// compare).  Rules on gfx950 (DESIGN 8, MI355X_MICROARCH "inter-workgroup visibility"): producer data leaves through write-through (sc1) stores, every
// storing wave drains vmcnt, one barrier, then ONE lane bumps the counters with relaxed agent-scope atomics; consumers poll relaxed (one lane per watched
// frame, s_sleep between polls) and read the activations with sc1 loads — no agent-scope fence anywhere (a release fence writes back the XCD's whole L2,
// i.e. the learner's working set).  Progress does not depend on placement or timing: a block only waits for blocks with LOWER ids, and the dispatcher hands out
// workgroups in id order, so the unfinished block with the lowest id is always running or next in line.  Should that order ever not hold, a bounded spin turns
// the deadlock into an error word the host checks (af_err).  The per-frame tail (split-K reduce + heads + sampling + env step) stays its own launch: its
// 57 KB of LDS would cap every stage of a common kernel at two blocks per CU.
// RESULT (round 6, one MI355X, nothing else on the GPU): bit-identical actions / log-probs / values (the parity and end-to-end suites run it), and
// 106-110 us per 120-env step against 51 for the five launches.  Timing builds (-DAF_ABL, results wrong on purpose): plain stores only 110, plain loads only
// 104, no waiting 91, all three 39 (= 29 for the four layers with nothing between them + the tail).  These small-batch blocks are LATENCY-bound — a chunk is 256
// cycles of MFMA per wave behind a two-chunk register prefetch — and what a kernel boundary gives them for free is exactly what coherent dataflow takes
// away: the im2col re-reads of a tile hit the CU's L1 (an sc1 load never does), and a write-through store drops its line from the XCD's L2, so every first
// touch of an activation is a fabric round trip, eight times over (the consumers of a frame sit on all eight XCDs).  The legal variant that keeps L1 / L2
// (plain stores + an agent-scope release per producer block) writes back the XCD's whole L2 per block — the learner's working set, round 5's finding.  Even
// the unreachable 39 us would be 24 % of the rollout.  Kept behind CBM_ACTOR_FUSED=1 as the measured experiment; the product runs the five launches.
static __device__ __forceinline__ float4 af_ld16_wt(__amdgpu_buffer_rsrc_t r, uint32_t float_off) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4u), 0, (AF_ABL & 2) ? 0 : 16 /* sc1: served by L2, never by this CU's L1 */);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// wave 0: lane l < n watches counter cnt[first + l]; returns (block-wide) once every watched counter has reached `target`
static __device__ __forceinline__ void af_wait(const uint32_t* cnt, int first, int n, uint32_t target, uint32_t* err) {
  if (AF_ABL & 4) return;
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    const uint32_t* p = cnt + first + (l < n ? l : 0);
    for (unsigned spins = 0;; ++spins) {
      const uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((int)(v - target) >= 0)) break;
      if (spins > (1u << 22)) { if (l == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }   // (~1 s: a producer that never ran)
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
}
// after the epilogue's write-through stores: rows [x0, x0 + BX) of a layer with RPF rows per frame touch at most two frames (BX <= RPF)
static __device__ __forceinline__ void af_signal(uint32_t* cnt, int x0, int BX, int M, int RPF) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its stores have left
  __syncthreads();
  if (threadIdx.x < 2) {
    const int hi = min(x0 + BX, M), f = x0 / RPF + (int)threadIdx.x;
    const int lo_f = max(x0, f * RPF), hi_f = min(hi, (f + 1) * RPF);
    if (hi_f > lo_f) __hip_atomic_fetch_add(cnt + f, (uint32_t)(hi_f - lo_f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// igemm_s16_kernel's body (ROWPTR_S16 path) for block (bxi, byi, z) of problem p; A_WT: the A operand was produced by this launch -> sc1 loads
template <class P, int BX, int BY, int BR, bool A_WT>
static __device__ __forceinline__ void af_stage(const P& p, float* smem, int bxi, int byi, int z, size_t a_bytes) {
  static_assert(BX % 16 == 0 && BY == 32 && BR % 32 == 0 && igemm_rowptr_s16<P>::value, "small-batch tile");
  constexpr int NT16 = (BX / 16) * (BY / 16) / 4;
  constexpr int PA = BR + 4, PB = BY, ASZ = BX * PA, BSZ = BR * PB;
  constexpr int NVA = (BX * BR / 4 + 255) / 256, NVB = (BR * BY / 4 + 255) / 256;
  float* As = smem;
  float* Bs = smem + 2 * ASZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  const int x0 = bxi * BX, y0 = byi * BY;
  int rlo, rhi;
  p.r_range(z, rlo, rhi);
  const int nchunk = (rhi - rlo + BR - 1) / BR;
  f32x4_mfma acc[NT16];
#pragma unroll
  for (int i = 0; i < NT16; ++i) acc[i] = f32x4_mfma{0.0f, 0.0f, 0.0f, 0.0f};
  __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.a_origin_wt(), 0, (int)a_bytes, 0x00020000);
  uint32_t arow[NVA], brow[NVB];
#pragma unroll
  for (int j = 0; j < NVA; ++j) { const int v = tid + 256 * j, rq = v % (BR / 4), xl = v / (BR / 4); arow[j] = p.a_off(x0 + xl, 4 * rq, 0); }
#pragma unroll
  for (int j = 0; j < NVB; ++j) { const int v = tid + 256 * j, yq = v % (BY / 4), rl = v / (BY / 4); brow[j] = p.b_off(rl, y0 + 4 * yq, 0); }
  auto gload = [&](int c, float4 (&ra)[NVA], float4 (&rb)[NVB]) {
    const int r0 = rlo + c * BR;
    const uint32_t ao = p.a_chunk(r0), bo = p.b_chunk(r0);
#pragma unroll
    for (int j = 0; j < NVA; ++j)
      if (BX * BR / 4 % 256 == 0 || tid + 256 * j < BX * BR / 4) { if constexpr (A_WT) ra[j] = af_ld16_wt(arsrc, arow[j] + ao); else ra[j] = p.rp_a(arow[j] + ao); }
#pragma unroll
    for (int j = 0; j < NVB; ++j)
      if (BR * BY / 4 % 256 == 0 || tid + 256 * j < BR * BY / 4) rb[j] = p.rp_b(brow[j] + bo);
  };
  auto sstore = [&](int buf, const float4 (&ra)[NVA], const float4 (&rb)[NVB]) {
    float* A_ = As + buf * ASZ;
    float* B_ = Bs + buf * BSZ;
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) { const int rq = v % (BR / 4), xl = v / (BR / 4); *reinterpret_cast<float4*>(A_ + xl * PA + 4 * rq) = ra[j]; }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) { const int yq = v % (BY / 4), rl = v / (BY / 4); *reinterpret_cast<float4*>(B_ + rl * PB + ((4 * yq) ^ (((rl ^ (rl >> 1)) & 1) << 4))) = rb[j]; }
    }
  };
  auto compute = [&](int buf) {
    const float* A_ = As + buf * ASZ;
    const float* B_ = Bs + buf * BSZ;
#pragma unroll
    for (int st = 0; st < BR / 4; ++st) {
#pragma unroll
      for (int i = 0; i < NT16; ++i) {
        const int q = wave * NT16 + i, tx = q / (BY / 16), ty = q % (BY / 16);
        const float a = A_[(tx * 16 + r16) * PA + 4 * st + g4];
        const float b = B_[(4 * st + g4) * PB + ((ty * 16 + r16) ^ (((g4 ^ (g4 >> 1)) & 1) << 4))];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
    }
  };
  float4 a0[NVA], b0[NVB], a1[NVA], b1[NVB];
  float bpre[NT16];
#pragma unroll
  for (int i = 0; i < NT16; ++i) { const int q = wave * NT16 + i, ty = q % (BY / 16); bpre[i] = igemm_bias_pre<P>::value ? p.bias_pre(y0 + ty * 16 + r16) : 0.0f; }
  gload(0, a0, b0);
  sstore(0, a0, b0);
  if (nchunk > 1) gload(1, a0, b0);
  __syncthreads();
  int buf = 0, c = 0;
  while (true) {
    if (c + 2 < nchunk) gload(c + 2, a1, b1);
    compute(buf);
    if (c + 1 < nchunk) sstore(buf ^ 1, a0, b0);
    __syncthreads();
    buf ^= 1;
    if (++c >= nchunk) break;
    if (c + 2 < nchunk) gload(c + 2, a0, b0);
    compute(buf);
    if (c + 1 < nchunk) sstore(buf ^ 1, a1, b1);
    __syncthreads();
    buf ^= 1;
    if (++c >= nchunk) break;
  }
#pragma unroll
  for (int i = 0; i < NT16; ++i) {
    const int q = wave * NT16 + i, tx = q / (BY / 16), ty = q % (BY / 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) p.store_wt(x0 + tx * 16 + 4 * g4 + e, y0 + ty * 16 + r16, acc[i][e], bpre[i], z);
  }
}
struct ActorFusedArgs {
  const uint8_t* obs; const float* P; int64_t w[4], b[4];
  float* act1; float* act2; float* act3; float* dense_part;
  int E, ksplit; uint32_t* cnt; uint32_t epoch; uint32_t* err; int n1, n2, n3, nd, maxB;
};
using AfC1 = Conv1Fwd<IgemmTile<64, 64, 16, 2, 2>>;
using AfC2 = ConvFwd<IgemmTile<64, 64, 16, 2, 2>, 4, 4, 2, 32, 64, 20, 20, 9, 9>;
using AfC3 = ConvFwd<IgemmTile<64, 64, 16, 2, 2>, 3, 3, 1, 64, 64, 9, 9, 7, 7>;
using AfD = DenseFwd<IgemmTile<64, 64, 16, 2, 2>, true>;
__global__ __launch_bounds__(256, 2) void actor_fused_kernel(const ActorFusedArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 64 * 36 + 2 * 32 * 32];     // the largest stage (conv1: 64-row tiles); ONE LDS object
  __builtin_amdgcn_s_setprio(3);   // (see igemm_s16_kernel)
  int b = blockIdx.x;
  const int E = a.E;
  uint32_t* c1 = a.cnt; uint32_t* c2 = a.cnt + a.maxB; uint32_t* c3 = a.cnt + 2 * a.maxB;
  if (b < a.n1) {                                   // conv1: uint8 frames -> act1 [E*400][32]
    const AfC1 p{a.obs, nullptr, a.P + a.w[0], a.P + a.b[0], a.act1, E * 400, nullptr};
    af_stage<AfC1, 64, 32, 32, false>(p, smem, b, 0, 0, 0);
    af_signal(c1, b * 64, 64, E * 400, 400);
    return;
  }
  b -= a.n1;
  if (b < a.n2) {                                   // conv2: act1 -> act2 [E*81][64], two column blocks per row tile
    const int bx = b >> 1, by = b & 1, x0 = bx * 32, M = E * 81;
    const int f0 = x0 / 81, f1 = (min(x0 + 32, M) - 1) / 81;
    af_wait(c1, f0, f1 - f0 + 1, a.epoch * 400u, a.err);
    const AfC2 p{a.act1, a.P + a.w[1], a.P + a.b[1], a.act2, M, nullptr};
    af_stage<AfC2, 32, 32, 32, true>(p, smem, bx, by, 0, (size_t)E * 12800 * 4);
    af_signal(c2, x0, 32, M, 81);
    return;
  }
  b -= a.n2;
  if (b < a.n3) {                                   // conv3: act2 -> act3 [E*49][64]
    const int bx = b >> 1, by = b & 1, x0 = bx * 32, M = E * 49;
    const int f0 = x0 / 49, f1 = (min(x0 + 32, M) - 1) / 49;
    af_wait(c2, f0, f1 - f0 + 1, a.epoch * 162u, a.err);
    const AfC3 p{a.act2, a.P + a.w[2], a.P + a.b[2], a.act3, M, nullptr};
    af_stage<AfC3, 32, 32, 32, true>(p, smem, bx, by, 0, (size_t)E * 5184 * 4);
    af_signal(c3, x0, 32, M, 49);
    return;
  }
  b -= a.n3;
  {                                                 // dense, split-K partials [ksplit][E][512]: x tile (32 frames) major, then K slice, then column block
    const int per_x = a.ksplit * 16, bx = b / per_x, r = b - bx * per_x, z = r >> 4, by = r & 15, x0 = bx * 32;
    af_wait(c3, x0, min(32, E - x0), a.epoch * 98u, a.err);
    const AfD p{a.act3, a.P + a.w[3], a.P + a.b[3], a.dense_part, E, 3136, 512, 3136 / a.ksplit};
    af_stage<AfD, 32, 32, 32, true>(p, smem, bx, by, z, (size_t)E * 3136 * 4);
  }
}
static bool actor_fused_ok(const NatureLayout& L, int B, int dense_ksplit, const NatureWs& ws) {
  // OFF by default: measured 2.1x SLOWER than the five launches (106-110 vs 51 us per 120-env step, profiles/r06_actor_dataflow.txt) — see the note below
  static const bool on = [] { const char* e = getenv("CBM_ACTOR_FUSED"); return e && e[0] == '1'; }();
  return on && ws.af_cnt && B <= ws.maxB && dense_ksplit > 1 && dense_ksplit <= 16 && 3136 % (dense_ksplit * 32) == 0 && L.A + 1 <= 32;
}
static void launch_actor_fused(const NatureLayout& L, const float* P, const uint8_t* obs, int B, int dense_ksplit, NatureWs& ws, hipStream_t st) {
  if (ws.af_err_host && *ws.af_err_host) { cbm_launch_fail("dataflow actor step: a block gave up waiting for its producers (workgroups not dispatched in id order?)"); return; }
  ActorFusedArgs a;
  a.obs = obs; a.P = P;
  for (int i = 0; i < 4; ++i) { a.w[i] = L.w[i]; a.b[i] = L.b[i]; }
  a.act1 = ws.act1; a.act2 = ws.act2; a.act3 = ws.act3; a.dense_part = ws.dense_part;
  a.E = B; a.ksplit = dense_ksplit; a.cnt = ws.af_cnt; a.epoch = ++ws.af_epoch; a.err = ws.af_err_dev; a.maxB = ws.maxB;
  a.n1 = (B * 400 + 63) / 64; a.n2 = 2 * ((B * 81 + 31) / 32); a.n3 = 2 * ((B * 49 + 31) / 32); a.nd = ((B + 31) / 32) * dense_ksplit * 16;
  hipLaunchKernelGGL(actor_fused_kernel, dim3(a.n1 + a.n2 + a.n3 + a.nd), dim3(256), 0, st, a);
}

static void launch_heads_fwd(const float* hid, const float* Wa, const float* ba, const float* Wc, const float* bc, int B, int A, int HD,
                             float* logits, float* value, hipStream_t st) {
  if (HD % 64 == 0 && A + 1 <= 32) {
    HeadsFwd p{hid, Wa, ba, Wc, bc, logits, value, B, A, HD};
    igemm_s16_launch<32, 32, 64>(p, 1, st);
    return;
  }
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((B + 7) / 8), dim3(256), (8 * HD + HD * (A + 1)) * sizeof(float), st, hid, Wa, ba, Wc, bc, B, A, HD,
                     logits, value);
}

// ------------------------------------------------------------------------------------------ backward
// heads dgrad: dhid[m][k] = (sum_j dzv[m][j] * Wac[k][j]) * (hid > 0), j over A logits + value.
// One thread per hidden unit k keeps its A+1 weights in registers (one pass over Wa / Wc per block of HG_FR frames instead of A strided loads
// per output) and walks HG_FR frames whose dzv rows sit in LDS (broadcast reads); hid / dhid accesses are coalesced along k.
#define HG_FR 4
template <int HD>
__global__ __launch_bounds__(HD) void heads_dgrad_kernel(const float* dzv, const float* Wa, const float* Wc, const float* hid, int B, int A, float* dhid) {
  __shared__ float ds[HG_FR][32];
  const int k = threadIdx.x, m0 = blockIdx.x * HG_FR;
  for (int i = k; i < HG_FR * 32; i += HD) { const int f = m0 + i / 32; ds[i / 32][i % 32] = f < B ? dzv[(size_t)f * 32 + i % 32] : 0.0f; }
  float w[28];
#pragma unroll
  for (int a = 0; a < 28; ++a) w[a] = a < A ? Wa[k * A + a] : 0.0f;
  __syncthreads();
  const float wc = Wc[k];
#pragma unroll 4
  for (int f = 0; f < HG_FR; ++f) {
    const int m = m0 + f;
    if (m >= B) break;
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 28; ++a) if (a < A) s = fmaf(ds[f][a], w[a], s);
    s = fmaf(ds[f][A], wc, s);
    const size_t i = (size_t)m * HD + k;
    dhid[i] = hid[i] > 0.0f ? s : 0.0f;
  }
}
// ---- heads forward + PPO loss + heads input gradient, one launch (PPO learner minibatches) ------------------------------------------------
// The three were separate launches of 16 + 8 + 14 us for a few MFLOP: hid [B][HD] read twice, logits / dzv round-tripped through HBM.  One block
// per 16 samples: (A) hid rows -> LDS, head weights straight from L2 into the registers of waves 0 / 1 (HeadsFwd's k-ascending MFMA chain: the
// same logits bits); (B) logits / value -> LDS; (C) the loss head of ppo_loss.h on 32 lanes per sample, dzv -> LDS + HBM (the heads weight
// gradient reads it), block partial of the four statistics; (D) dhid[m][k] = (sum_j dzv[m][j] Wac[k][j]) * (hid > 0) as 16x16x4 MFMAs over
// j (5 steps for A = 18), eight 16-column tiles per wave, the weight fragments requested before the loss math.
template <int HD>
__global__ __launch_bounds__(256) void ppo_heads_fused_kernel(const float* hid, const float* Wa, const float* ba, const float* Wc, const float* bc, int B, int A,
                                                              const int32_t* idx, const int32_t* actions, const float* old_logprob, const float* adv,
                                                              const float* target, float clip_coef, float ent_coef, float vf_coef, float* logits_out,
                                                              float* value_out, float* dzv, float* dhid, float* partials) {
  constexpr int PH = HD + 4, NT = HD / 64;         // NT: 16-column dhid tiles per wave
  __shared__ __attribute__((aligned(16))) float hs[16 * PH];    // hid rows
  __shared__ __attribute__((aligned(16))) float wl[HD * 31];    // [HD][A] actor weights as they lie in memory (A + 1 <= 32)
  __shared__ float wc[HD];                                      // critic weights
  __shared__ float lg[16][33];
  __shared__ float dz[16][36];
  __shared__ float red[16][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g4 = lane >> 4;
  const int m0 = blockIdx.x * 16;
  TT(0);
  // (A) hid rows and both weight matrices -> LDS by the load unit (global_load_lds: 1 KB per wave instruction, no staging registers, no ds_write):
  // a hid row is two such copies behind its padded row base, the actor matrix HD / 256 * A of them as it lies in memory, the critic column HD / 64
  // 256-byte ones.  (Through registers this phase and the chain below were 31 us for a block that owns a CU alone — the same pattern the actor tail
  // had: profiles/NOTES_r03_r04.md round 4.)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // the samples' scalars are a gather through idx — two dependent round trips: the index is requested first, the scalars once the staging copies are
  // on their way (vmcnt retires in order: waiting for the index does not wait for the copies behind it)
  int n_gather = 0;
  if (tid < 16) { const int ii = min(m0 + tid, B - 1); n_gather = idx ? idx[ii] : ii; }
  for (int q = wave_u; q < 16 * (HD / 256); q += 4) {
    const int row = q / (HD / 256), part = q % (HD / 256);
    ig_glds16(hid + (size_t)min(m0 + row, B - 1) * HD + part * 256 + lane * 4, hs + row * PH + part * 256);
  }
  if (((uintptr_t)Wa & 15) == 0) {
    for (int q = wave_u; q < (HD / 256) * A; q += 4) ig_glds16(Wa + (size_t)(64 * q + lane) * 4, wl + 64 * q * 4);
  } else {
    for (int i = tid; i < HD * A; i += 256) wl[i] = Wa[i];
  }
  for (int q = wave_u; q < HD / 64; q += 4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wc + 64 * q + lane), (__attribute__((address_space(3))) void*)(wc + 64 * q), 4, 0, 0);
  __shared__ int s_act[16];
  __shared__ float s_olp[16], s_adv[16], s_tgt[16];
  if (tid < 16) { const int n = n_gather; s_act[tid] = actions[n]; s_olp[tid] = old_logprob[n]; s_adv[tid] = adv[n]; s_tgt[tid] = target[n]; }
  TT(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TT(2);
  {
    const int n = wave * 16 + r16;
    if (wave < 2) {
      // every fragment of the 128-step chain comes out of LDS BEFORE the first MFMA, through running pointers (read inside the chain each group of
      // MFMAs waits for its own LDS round trip; indexed as base[st * stride] each read costs a quarter-rate integer multiply).  Columns past A + 1
      // multiply whatever the critic column holds: their results are never stored.
      const float* wq = (n < A ? wl + n : wc) + g4 * (n < A ? A : 1);
      const int wstep = 4 * (n < A ? A : 1);
      float bw[HD / 4], av[HD / 4];
      {
        const float* wr = wq;
        const float* ar = hs + r16 * PH + g4;
#pragma unroll
        for (int st = 0; st < HD / 4; ++st) { bw[st] = *wr; wr += wstep; av[st] = ar[4 * st]; }
      }
      __builtin_amdgcn_sched_barrier(0);
      TT(3);
      f32x4_mfma acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int st = 0; st < HD / 4; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], bw[st], acc, 0, 0, 0);
      TT(4);
      const float bias = n < A ? ba[n] : (n == A ? bc[0] : 0.0f);
      if (n <= A) {
#pragma unroll
        for (int e = 0; e < 4; ++e) lg[4 * g4 + e][n] = acc[e] + bias;
      }
    }
  }
  const int nst = (A + 4) / 4;                      // j runs over A logits + the value column
  __syncthreads();
  TT(5);
  const float invN = 1.0f / (float)B;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = pass * 8 + (tid >> 5), i = m0 + row, j = tid & 31;
    const bool live = i < B;
    const float zj = lg[row][j < A ? j : A - 1], val = lg[row][A];
    PpoSampleStats ss;
    ss.pg = ss.dv2 = ss.ent = ss.kl = 0.0f;
    const float d = ppo_loss_lane(zj, j, A, s_act[row], val, s_olp[row], s_adv[row], s_tgt[row], clip_coef, ent_coef, vf_coef, invN, ss);
    dz[row][j] = live ? d : 0.0f;
    if (live) {
      dzv[(size_t)i * 32 + j] = d;
      if (j < A) logits_out[(size_t)i * A + j] = zj;
      else if (j == A) value_out[i] = val;
    }
    if (j == 0) { red[row][0] = live ? ss.pg : 0.0f; red[row][1] = live ? ss.dv2 : 0.0f; red[row][2] = live ? ss.ent : 0.0f; red[row][3] = live ? ss.kl : 0.0f; }
  }
  TT(6);
  __syncthreads();
  TT(7);
  if (tid < 4) {
    float r[16], v = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) r[q] = red[q][tid];
#pragma unroll
    for (int q = 0; q < 16; ++q) v += r[q];
    partials[blockIdx.x * 4 + tid] = v;
  }
  // dhid = dz x [Wa | Wc]^T, 16-column tiles.  No guard is a branch: all eight k steps run (A + 1 <= 29 columns; dz is zero past column A, the weight
  // operand is selected to zero there, and acc + 0 * 0 leaves acc as it is), the weights are read wherever the index lands (inside wl) and selected
  // afterwards, the dz fragments are read once — guarded per step the compiler serialised 40 branch / LDS read / wait / MFMA groups: 7.9 us of this
  // kernel's 21.5 (tools/heads_trace.py)
  (void)nst;
  const int Av = cbm_opaque_vgpr(A);
  float az[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) az[st] = dz[r16][4 * st + g4];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int k = (wave * NT + ti) * 16 + r16;
    const float wck = wc[k];
    float bv[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) { const int jc = 4 * st + g4; const float x = wl[k * A + jc]; bv[st] = jc < Av ? x : (jc == Av ? wck : 0.0f); }
    f32x4_mfma acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int st = 0; st < 8; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(az[st], bv[st], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 4 * g4 + e, m = m0 + row;
      if (m < B) dhid[(size_t)m * HD + k] = hs[row * PH + k] > 0.0f ? acc[e] : 0.0f;
    }
  }
  TT(8);
}
bool ppo_heads_fusable(const NatureLayout& L) { return L.A + 1 <= 32 && (L.hid == 512 || L.hid == 256); }
void launch_ppo_heads_fused(const NatureLayout& L, const float* P, NatureWs& ws, int B, const int32_t* idx, const int32_t* actions,
                            const float* old_logprob, const float* adv, const float* target, float clip_coef, float ent_coef, float vf_coef,
                            float* partials, float* stats5, hipStream_t st) {
  const int nblk = (B + 15) / 16;
  if (L.hid == 512)
    hipLaunchKernelGGL(ppo_heads_fused_kernel<512>, dim3(nblk), dim3(256), 0, st, ws.hid, P + L.w[4], P + L.b[4], P + L.w[5], P + L.b[5], B, L.A, idx, actions,
                       old_logprob, adv, target, clip_coef, ent_coef, vf_coef, ws.logits, ws.value, ws.dzv, ws.dhid, partials);
  else
    hipLaunchKernelGGL(ppo_heads_fused_kernel<256>, dim3(nblk), dim3(256), 0, st, ws.hid, P + L.w[4], P + L.b[4], P + L.w[5], P + L.b[5], B, L.A, idx, actions,
                       old_logprob, adv, target, clip_coef, ent_coef, vf_coef, ws.logits, ws.value, ws.dzv, ws.dhid, partials);
  // the statistics are summed by the backward pass's reduction launch (RedBatch::add_stats); a caller without a backward pass flushes them itself
  ws.pending_stats = PendingStats{partials, stats5, nblk, B, ent_coef, vf_coef};
}
void flush_pending_stats(NatureWs& ws, hipStream_t st) {
  const PendingStats ps = ws.pending_stats;
  if (!ps.partials) return;
  ws.pending_stats = PendingStats{};
  launch_ppo_stats(ps.partials, ps.nblk, ps.N, ps.ent_coef, ps.vf_coef, ps.stats5, st);
}

static void launch_heads_dgrad(const float* dzv, const float* Wa, const float* Wc, const float* hid, int B, int A, int HD, float* dhid, hipStream_t st) {
  const int nb = (B + HG_FR - 1) / HG_FR;
  switch (HD) {   // one thread per hidden unit: the widths a context can be created with (`--hiddens`, multiples of 64 up to 512)
    case 512: hipLaunchKernelGGL(heads_dgrad_kernel<512>, dim3(nb), dim3(512), 0, st, dzv, Wa, Wc, hid, B, A, dhid); break;
    case 448: hipLaunchKernelGGL(heads_dgrad_kernel<448>, dim3(nb), dim3(448), 0, st, dzv, Wa, Wc, hid, B, A, dhid); break;
    case 384: hipLaunchKernelGGL(heads_dgrad_kernel<384>, dim3(nb), dim3(384), 0, st, dzv, Wa, Wc, hid, B, A, dhid); break;
    case 320: hipLaunchKernelGGL(heads_dgrad_kernel<320>, dim3(nb), dim3(320), 0, st, dzv, Wa, Wc, hid, B, A, dhid); break;
    case 256: hipLaunchKernelGGL(heads_dgrad_kernel<256>, dim3(nb), dim3(256), 0, st, dzv, Wa, Wc, hid, B, A, dhid); break;
    case 192: hipLaunchKernelGGL(heads_dgrad_kernel<192>, dim3(nb), dim3(192), 0, st, dzv, Wa, Wc, hid, B, A, dhid); break;
    case 128: hipLaunchKernelGGL(heads_dgrad_kernel<128>, dim3(nb), dim3(128), 0, st, dzv, Wa, Wc, hid, B, A, dhid); break;
    default: hipLaunchKernelGGL(heads_dgrad_kernel<64>, dim3(nb), dim3(64), 0, st, dzv, Wa, Wc, hid, B, A, dhid); break;
  }
}

// dense dgrad: dact3[m][j] = sum_n dhid[m][n] * Wd[j][n]; stored masked into the zero-bordered
// dact3pad [S][11][11][64] (data at rows/cols 2..8).
template <class TileT>
struct DenseDgrad {
  using Tile = TileT;
  static constexpr bool A_RX = false, B_YR = true, BIAS_GRAD = false;
  static constexpr int NCLS = 1;
  const float* dhid; const float* Wd; const float* act3; float* dact3pad; int M;
  __host__ __device__ int X() const { return M; }
  __host__ __device__ int Y() const { return 3136; }
  __device__ void r_range(int, int& lo, int& hi) const { lo = 0; hi = 512; }
  __device__ float4 load_a(int m, int r, int, int) const {
    return *reinterpret_cast<const float4*>(dhid + (size_t)min(m, M - 1) * 512 + r);
  }
  __device__ float4 load_b(int r, int y, int, int) const {   // B[r=n..n+3][y=j] = Wd[j][n..n+3]
    return *reinterpret_cast<const float4*>(Wd + (size_t)min(y, 3135) * 512 + r);
  }
  __device__ void store(int m, int j, float v, int, int) const {
    if (m >= M || j >= 3136) return;
    const int pos = j >> 6, c = j & 63, hh = pos / 7, ww = pos - hh * 7;
    const bool on = act3[(size_t)m * 3136 + j] > 0.0f;
    dact3pad[((size_t)(m * 11 + hh + 2) * 11 + ww + 2) * 64 + c] = on ? v : 0.0f;
  }
  // bit-mask epilogue (igemm.h BITMASK): the ReLU mask of act3 comes as one word per (frame, 32 flatten columns)
  static constexpr bool BITMASK = true;
  const uint32_t* mask;
  __device__ uint32_t mask_word(int m, int j32, int) const { return mask[(size_t)min(m, M - 1) * 98 + (min(j32, 3135) >> 5)]; }
  __device__ void store_on(int m, int j, float v, bool on, int, int) const {
    if (m >= M || j >= 3136) return;
    const int pos = j >> 6, c = j & 63, hh = pos / 7, ww = pos - hh * 7;
    dact3pad[((size_t)(m * 11 + hh + 2) * 11 + ww + 2) * 64 + c] = on ? v : 0.0f;
  }
  static constexpr bool ROWEPI = true;   // consecutive frames m are 121*64 floats apart in dact3pad
  __device__ EpiRow epi_row(int m0, int j, int) const {
    const int jj = min(j, 3135), pos = jj >> 6, c = jj & 63, hh = pos / 7, ww = pos - hh * 7;
    return EpiRow{dact3pad + ((size_t)(m0 * 11 + hh + 2) * 11 + ww + 2) * 64 + c, j < 3136 ? M - m0 : 0, 7744u};
  }
};

// conv3 dgrad (3x3 s1), position-major: reads dact3pad, writes masked into dact2pad [S][11][11][64] (data at 1..9); r = (jh, jw, co) over the
// flipped taps.  A frame-major GEMM (x = (frame, 9x9 pixel)) multiplies all 9 taps for every output pixel although the 7x7 dY sits in a
// zero border — only 49/81 of its flops touch data.  Here an x-tile is ONE output pixel (ih, iw) of BX consecutive frames, so the taps
// that can be non-zero are the same for all its rows (jh in [max(0,2-ih), min(2,8-ih)], same for jw: 1, 2 or 3 per axis) and the block
// reduces over exactly those (igemm.h KSKIP): 441 instead of 729 tap-tiles per frame tile.  Tiles are ordered frame-tile-major, pixel-minor, so
// the 81 blocks that read the same BX frames of dY run back to back on one XCD.  Rows beyond S (last frame tile) load frame S-1 and store nothing.
template <class TileT>
struct Conv3DgradPos {
  using Tile = TileT;
  static constexpr bool A_RX = false, B_YR = true, BIAS_GRAD = false, KSKIP = true;
  static constexpr int NCLS = 1;
  const float* dypad; const float* W; const float* act2; float* dxpad; int S;
  const int32_t* order;   // x-tile index (after the kernel's XCD remap) -> frame_tile * 81 + pixel
  __host__ __device__ int X() const { return ((S + Tile::BX - 1) / Tile::BX) * 81 * Tile::BX; }
  __host__ __device__ int Y() const { return 64; }
  __device__ void r_range(int, int& lo, int& hi) const { lo = 0; hi = 576; }
  __device__ int block_ctx(int x0, int) const {
    const int p = order[x0 / Tile::BX] % 81, ih = p / 9, iw = p - ih * 9;
    const int jh0 = max(0, 2 - ih), jh1 = min(2, 8 - ih), jw0 = max(0, 2 - iw), jw1 = min(2, 8 - iw);
    const int njw = jw1 - jw0 + 1, nt = (jh1 - jh0 + 1) * njw;
    return jh0 | (jw0 << 2) | (njw << 4) | (nt << 8);
  }
  __device__ int block_k(int ctx) const { return (ctx >> 8) * 64; }
  __device__ int r_map(int ctx, int rc) const {
    const int t = rc >> 6, njw = (ctx >> 4) & 15, q = t / njw;
    return ((ctx & 3) + q) * 192 + (((ctx >> 2) & 3) + (t - q * njw)) * 64 + (rc & 63);
  }
  __device__ void decode(int x, int& s, int& ih, int& iw) const {
    const int xt = x / Tile::BX, tile = order[xt], t = tile / 81, p = tile - t * 81;
    s = t * Tile::BX + (x - xt * Tile::BX); ih = p / 9; iw = p - ih * 9;
  }
  __device__ float4 load_a(int x, int r, int, int) const {
    int s, ih, iw;
    decode(x, s, ih, iw);
    s = min(s, S - 1);
    const int jh = r / 192, rem = r - jh * 192;
    return *reinterpret_cast<const float4*>(dypad + ((size_t)(s * 11 + ih + jh) * 11 + iw) * 64 + rem);
  }
  __device__ float4 load_b(int r, int ci, int, int) const {
    const int jh = r / 192, rem = r - jh * 192, jw = rem >> 6, co = rem & 63;
    return *reinterpret_cast<const float4*>(W + ((size_t)((2 - jh) * 3 + (2 - jw)) * 64 + ci) * 64 + co);
  }
  static constexpr bool ROWPTR = 64 % TileT::BR == 0;   // a chunk stays inside one 64-wide tap; 189 -> 168 us
  __device__ const float* a_origin() const { return dypad; }
  __device__ const float* b_origin() const { return W; }
  __device__ uint32_t a_off(int x, int rl, int) const {
    int s, ih, iw;
    decode(x, s, ih, iw);
    s = min(s, S - 1);
    return (uint32_t)(((s * 11 + ih) * 11 + iw) * 64 + rl);
  }
  __device__ uint32_t a_chunk(int r0) const { const int jh = r0 / 192; return (uint32_t)(jh * 11 * 64 + (r0 - jh * 192)); }
  __device__ uint32_t b_off(int rl, int ci, int) const { return (uint32_t)(ci * 64 + rl); }
  __device__ uint32_t b_chunk(int r0) const {
    const int jh = r0 / 192, rem = r0 - jh * 192, jw = rem >> 6;
    return (uint32_t)(((2 - jh) * 3 + (2 - jw)) * 4096 + (rem & 63));
  }
  __device__ void store(int x, int ci, float v, int, int) const {
    int s, ih, iw;
    decode(x, s, ih, iw);
    if (s >= S) return;
    const bool on = act2[((size_t)s * 81 + ih * 9 + iw) * 64 + ci] > 0.0f;
    dxpad[((size_t)(s * 11 + ih + 1) * 11 + iw + 1) * 64 + ci] = on ? v : 0.0f;
  }
  static constexpr bool BITMASK = true;
  const uint32_t* mask;   // act2 ReLU bits [S*81][2]
  __device__ uint32_t mask_word(int x, int ci32, int) const {
    int s, ih, iw;
    decode(x, s, ih, iw);
    return mask[((size_t)min(s, S - 1) * 81 + ih * 9 + iw) * 2 + (ci32 >> 5)];
  }
  __device__ void store_on(int x, int ci, float v, bool on, int, int) const {
    int s, ih, iw;
    decode(x, s, ih, iw);
    if (s >= S) return;
    dxpad[((size_t)(s * 11 + ih + 1) * 11 + iw + 1) * 64 + ci] = on ? v : 0.0f;
  }
  static constexpr bool ROWEPI = true;   // the rows of an x-tile are consecutive frames at one pixel: 121*64 floats apart in dact2pad
  __device__ EpiRow epi_row(int xt0, int ci, int) const {
    int s, ih, iw;
    decode(xt0, s, ih, iw);
    return EpiRow{dxpad + ((size_t)(s * 11 + ih + 1) * 11 + iw + 1) * 64 + ci, S - s, 7744u};
  }
};

// conv2 dgrad (4x4 s2): reads dact2pad, writes masked dact1.  The stride-2 transposed conv splits into 4 parity classes (ph, pw) of the input
// pixel, each a 2x2-tap VALID correlation: r = (jh, jw, co) with kh = ph + 2(1-jh), kw = pw + 2(1-jw).  Position-major AND class-merged: an
// x-tile is ONE half-resolution pixel (ihh, iwh) of BX consecutive frames — so the taps that fall on dY's zero border are the same for all
// its rows and are skipped per block (igemm.h KSKIP: ihh = 0 keeps only jh = 1, ihh = 9 only jh = 0, likewise iwh; exactly the algorithmic
// MACs) — and the four classes, which gather IDENTICAL rows of dY, are one GEMM with y = cls*32 + ci (N = 128).
template <class TileT>
struct Conv2DgradMergedPos {
  using Tile = TileT;
  static constexpr bool A_RX = false, B_YR = true, BIAS_GRAD = false, BITMASK = true, KSKIP = true;
  static constexpr int NCLS = 1;
  const float* dypad; const float* W; float* dact1; int S; const uint32_t* mask;
  __host__ __device__ int X() const { return ((S + Tile::BX - 1) / Tile::BX) * 100 * Tile::BX; }
  __host__ __device__ int Y() const { return 128; }
  __device__ void r_range(int, int& lo, int& hi) const { lo = 0; hi = 256; }
  __device__ int block_ctx(int x0, int) const {
    const int p = (x0 / Tile::BX) % 100, ihh = p / 10, iwh = p - ihh * 10;
    const int jh0 = ihh == 0 ? 1 : 0, jh1 = ihh == 9 ? 0 : 1, jw0 = iwh == 0 ? 1 : 0, jw1 = iwh == 9 ? 0 : 1;
    const int njw = jw1 - jw0 + 1, nt = (jh1 - jh0 + 1) * njw;
    return jh0 | (jw0 << 2) | (njw << 4) | (nt << 8);
  }
  __device__ int block_k(int ctx) const { return (ctx >> 8) * 64; }
  __device__ int r_map(int ctx, int rc) const {
    const int t = rc >> 6, njw = (ctx >> 4) & 15, q = t / njw;
    return ((ctx & 3) + q) * 128 + (((ctx >> 2) & 3) + (t - q * njw)) * 64 + (rc & 63);
  }
  __device__ void decode(int x, int& s, int& ihh, int& iwh) const {
    const int tile = x / Tile::BX, t = tile / 100, p = tile - t * 100;
    s = t * Tile::BX + (x - tile * Tile::BX); ihh = p / 10; iwh = p - ihh * 10;
  }
  __device__ float4 load_a(int x, int r, int, int) const {
    int s, ihh, iwh;
    decode(x, s, ihh, iwh);
    s = min(s, S - 1);
    const int jh = r >> 7, rem = r & 127;
    return *reinterpret_cast<const float4*>(dypad + ((size_t)(s * 11 + ihh + jh) * 11 + iwh) * 64 + rem);
  }
  __device__ float4 load_b(int r, int y, int, int) const {
    const int cls = y >> 5, ci = y & 31;
    const int jh = r >> 7, jw = (r >> 6) & 1, co = r & 63;
    const int kh = (cls >> 1) + 2 * (1 - jh), kw = (cls & 1) + 2 * (1 - jw);
    return *reinterpret_cast<const float4*>(W + ((size_t)(kh * 4 + kw) * 32 + ci) * 64 + co);
  }
  // row / chunk split of the same addresses (igemm.h ROWPTR; a 16-wide chunk never leaves one 64-wide tap): the generic load_a above decodes
  // (frame, ihh, iwh) with two integer divisions for every 16-byte load of every chunk — 4.46 VALU instructions per MFMA in round 5's counters
  static constexpr bool ROWPTR = 64 % TileT::BR == 0;
  __device__ const float* a_origin() const { return dypad; }
  __device__ const float* b_origin() const { return W; }
  __device__ uint32_t a_off(int x, int rl, int) const {
    int s, ihh, iwh;
    decode(x, s, ihh, iwh);
    s = min(s, S - 1);
    return (uint32_t)(((s * 11 + ihh) * 11 + iwh) * 64 + rl);
  }
  __device__ uint32_t a_chunk(int r0) const { return (uint32_t)((r0 >> 7) * (11 * 64) + (r0 & 127)); }
  __device__ uint32_t b_off(int rl, int y, int) const {        // B[r .. r+3][y] = W[kh][kw][ci][co .. co+3]: the class part of (kh, kw) and ci belong to the row
    const int cls = y >> 5, ci = y & 31;
    return (uint32_t)((((cls >> 1) * 4 + (cls & 1)) * 32 + ci) * 64 + rl);
  }
  __device__ uint32_t b_chunk(int r0) const {                  // the tap part of (kh, kw) = (2 (1 - jh), 2 (1 - jw)) and the chunk's first co
    const int jh = r0 >> 7, jw = (r0 >> 6) & 1;
    return (uint32_t)(((2 * (1 - jh)) * 4 + 2 * (1 - jw)) * 32 * 64 + (r0 & 63));
  }
  __device__ size_t pixel(int x, int cls, bool& ok) const {
    int s, ihh, iwh;
    decode(x, s, ihh, iwh);
    ok = s < S;
    return (size_t)(min(s, S - 1) * 20 + 2 * ihh + (cls >> 1)) * 20 + 2 * iwh + (cls & 1);
  }
  __device__ uint32_t mask_word(int x, int y32, int) const { bool ok; return mask[pixel(x, y32 >> 5, ok)]; }
  __device__ void store_on(int x, int y, float v, bool on, int, int) const {
    bool ok;
    const size_t px = pixel(x, y >> 5, ok);
    if (ok) dact1[px * 32 + (y & 31)] = on ? v : 0.0f;
  }
  __device__ void store(int x, int y, float v, int, int) const { store_on(x, y, v, true, 0, 0); }
  static constexpr bool ROWEPI = true;   // the rows of an x-tile are consecutive frames at one half-resolution pixel: 400*32 floats apart in dact1
  __device__ EpiRow epi_row(int xt0, int y, int) const {
    int s, ihh, iwh;
    decode(xt0, s, ihh, iwh);
    const int cls = y >> 5;
    return EpiRow{dact1 + ((size_t)(s * 20 + 2 * ihh + (cls >> 1)) * 20 + 2 * iwh + (cls & 1)) * 32 + (y & 31), S - s, 12800u};
  }
};

// ---- weight gradients: C[x = k][y = co] = sum_{r = m} A[m][k] * dY[m][co], split over r into
// partials [z][X][Y] (+ bias partial [z][Y]) reduced in ascending z (deterministic, ppo:30).
template <class TileT, int KH, int KW, int ST, int CI, int CO, int IH, int IW, int OH, int OW, int PADO>
struct ConvWgrad {  // dY lives in a zero-bordered [S][OH+2*PADO... = 11][11][CO] buffer
  using Tile = TileT;
  static constexpr bool A_RX = true, B_YR = false, BIAS_GRAD = true;
  static constexpr int NCLS = 1;
  static constexpr int KX = KH * KW * CI;
  const float* in; const float* dypad; float* part; float* bpart; int M, rps;
  __host__ __device__ int X() const { return KX; }
  __host__ __device__ int Y() const { return CO; }
  __device__ void r_range(int z, int& lo, int& hi) const { lo = z * rps; hi = min(lo + rps, M); }
  __device__ float4 load_a(int k, int m, int rhi, int) const {
    const bool ok = m < rhi && k < KX;
    m = min(m, rhi - 1); k = min(k, KX - 4);
    const int s = m / (OH * OW), p = m - s * (OH * OW), oh = p / OW, ow = p - oh * OW;
    const int kh = k / (KW * CI), rem = k - kh * (KW * CI);
    return f4sel(ok, *reinterpret_cast<const float4*>(in + ((size_t)(s * IH + oh * ST + kh) * IW + ow * ST) * CI + rem));
  }
  __device__ float4 load_b(int m, int y, int rhi, int) const {
    const bool ok = m < rhi;
    m = min(m, rhi - 1);
    const int s = m / (OH * OW), p = m - s * (OH * OW), oh = p / OW, ow = p - oh * OW;
    return f4sel(ok, *reinterpret_cast<const float4*>(dypad + ((size_t)(s * 11 + oh + PADO) * 11 + ow + PADO) * CO + y));
  }
  __device__ void store(int k, int y, float v, int z, int) const {
    if (k < KX) part[((size_t)z * KX + k) * CO + y] = v;
  }
  __device__ void store_bias(int y, float v, int z) const { bpart[z * CO + y] = v; }
};

// plain wgrad: C[x][y] = sum_m A[m][x] * G[m][y]   (dense: A = act3, G = dhid; heads: A = hid, G = dzv)
template <class TileT, bool PRE_RELU = false>
struct MatWgrad {
  using Tile = TileT;
  static constexpr bool A_RX = true, B_YR = false, BIAS_GRAD = true;
  static constexpr int NCLS = 1;
  const float* A; const float* G; float* part; float* bpart; int M, XK, YN, ldg, rps;
  __host__ __device__ int X() const { return XK; }
  __host__ __device__ int Y() const { return YN; }
  __device__ void r_range(int z, int& lo, int& hi) const { lo = z * rps; hi = min(lo + rps, M); }
  __device__ float4 load_a(int k, int m, int rhi, int) const {
    const bool ok = m < rhi && k < XK;
    const float4 v = f4sel(ok, *reinterpret_cast<const float4*>(A + (size_t)min(m, rhi - 1) * XK + min(k, XK - 4)));
    return PRE_RELU ? f4relu(v) : v;
  }
  __device__ float4 load_b(int m, int y, int rhi, int) const {
    const bool ok = m < rhi && y < YN;
    return f4sel(ok, *reinterpret_cast<const float4*>(G + (size_t)min(m, rhi - 1) * ldg + min(y, YN - 4)));
  }
  __device__ void store(int k, int y, float v, int z, int) const {
    if (k < XK && y < YN) part[((size_t)z * XK + k) * YN + y] = v;
  }
  __device__ void store_bias(int y, float v, int z) const { if (y < YN) bpart[z * YN + y] = v; }
};

// partial reduce: out[i] = sum_z part[z][i], fixed order (deterministic, ppo:30): the block's ZG z-groups each sum
// a strided subset of z ascending, then the ZG partials are added in ascending group order.
// mode 0: identity, 1: conv1 k=(c,kh,kw) -> HWIO, 2: heads [512][32] -> actor.w [512][A] / critic.w [512],
// 3: heads bias [32] -> actor.b [A] / critic.b [1]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, int nz, int XY, int Ycols, int mode, int A, int zg, float scale,
                                                            float* gw, float* gw2) {
  __shared__ float red[256];
  const int ow = 256 / zg;
  const int o = threadIdx.x % ow, g = threadIdx.x / ow;
  const int i = blockIdx.x * ow + o;
  float s = 0.0f;
  if (i < XY)
    for (int z = g; z < nz; z += zg) s += part[(size_t)z * XY + i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (g != 0 || i >= XY) return;
  for (int q = 1; q < zg; ++q) s += red[q * ow + o];
  s *= scale;
  const int x = i / Ycols, y = i - x * Ycols;
  if (mode == 0) gw[i] = s;
  else if (mode == 1) { const int c = x >> 6, kh = (x >> 3) & 7, kw = x & 7; gw[((kh * 8 + kw) * 4 + c) * 32 + y] = s; }
  else if (mode == 2) { if (y < A) gw[x * A + y] = s; else if (y == A) gw2[x] = s; }
  else { if (y < A) gw[y] = s; else if (y == A) gw2[0] = s; }
}
static void launch_reduce(const float* part, int nz, int XY, int Ycols, int mode, int A, float* gw, float* gw2, hipStream_t st, float scale = 1.0f) {
  const int zg = nz >= 64 ? 16 : (nz >= 8 ? 4 : 1);
  const int ow = 256 / zg;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((XY + ow - 1) / ow), dim3(256), 0, st, part, nz, XY, Ycols, mode, A, zg, scale, gw, gw2);
}

// The backward pass has ten of these reductions (weights and biases of five layers).  As ten launches they cost ~86 us per minibatch on
// the learner stream (most of them a few microseconds of work behind a launch); the partials of every layer now live in their own region
// and the reductions run as TWO launches: {heads, dense} right after the dense weight gradient (their result is the tail the data-parallel
// all-reduce waits for) and {conv3, conv2, conv1} at the end.  Same per-output summation order as the single launches -> same bits.
struct RedJob { const float* part; float* gw; float* gw2; int nz, XY, Ycols, mode, zg, block0; float scale; };
#define RED_MAX_JOBS 36   // Nature: 4 + 6 jobs; ResNet: 4 (dense + heads) + 30 (15 convs x {weights, bias}) — two launches beside an all-reduce, one without; + 1: the PPO loss statistics
struct RedJobs { RedJob j[RED_MAX_JOBS]; int n, A; float ent_coef, vf_coef; };
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const RedJobs jobs) {
  // four consecutive outputs per thread (16-byte loads; every XY is a multiple of 32), same z order per output as the scalar form
  __shared__ float4 red[256];
  int q = 0;
#pragma unroll
  for (int k = 1; k < RED_MAX_JOBS; ++k) if (k < jobs.n && (int)blockIdx.x >= jobs.j[k].block0) q = k;
  const RedJob& J = jobs.j[q];
  if (J.mode == 4) {   // the PPO loss statistics of this minibatch (one block; ppo_stats_kernel's arithmetic in its order, on wave 0): see RedBatch::add_stats
    if (threadIdx.x < 64) ppo_stats_wave(J.part, J.nz, J.XY, jobs.ent_coef, jobs.vf_coef, J.gw);
    return;
  }
  const int zg = J.zg, ow = 256 / zg, nz = J.nz, XY = J.XY;
  const int o = threadIdx.x % ow, g = threadIdx.x / ow;
  const int i = (((int)blockIdx.x - J.block0) * ow + o) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < XY)
    for (int z0 = g; z0 < nz; z0 += 8 * zg) {   // eight slices in flight (one after the other, the 5 slices of the dense layer were 5 HBM round trips per block)
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int z = z0 + u * zg;
        if (z < nz) v[u] = *reinterpret_cast<const float4*>(J.part + (size_t)z * XY + i);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (z0 + u * zg < nz) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }   // same z order as the scalar form
    }
  red[threadIdx.x] = s;
  __syncthreads();
  if (g != 0 || i >= XY) return;
  for (int r = 1; r < zg; ++r) { const float4 v = red[r * ow + o]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  const float out[4] = {s.x * J.scale, s.y * J.scale, s.z * J.scale, s.w * J.scale};
  const int A = jobs.A;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ii = i + c, x = ii / J.Ycols, y = ii - x * J.Ycols;
    const float v = out[c];
    if (J.mode == 0) J.gw[ii] = v;
    else if (J.mode == 1) { const int cc = x >> 6, kh = (x >> 3) & 7, kw = x & 7; J.gw[((kh * 8 + kw) * 4 + cc) * 32 + y] = v; }
    else if (J.mode == 2) { if (y < A) J.gw[x * A + y] = v; else if (y == A) J.gw2[x] = v; }
    else { if (y < A) J.gw[y] = v; else if (y == A) J.gw2[0] = v; }
  }
}
struct RedBatch {
  RedJobs jobs;
  int blocks = 0;
  // (the job table is a kernel argument of fixed size; a layer list that outgrows it fails the launch check instead of writing past it)
  bool room(int more) {
    if (jobs.n + more <= RED_MAX_JOBS) return true;
    cbm_launch_fail("reduction launch: %d jobs > %d", jobs.n + more, RED_MAX_JOBS);
    return false;
  }
  explicit RedBatch(int A) { jobs.n = 0; jobs.A = A; jobs.ent_coef = jobs.vf_coef = 0.0f; }
  void add(const float* part, int nz, int XY, int Ycols, int mode, float* gw, float* gw2, float scale = 1.0f) {
    const int zg = nz >= 64 ? 16 : (nz >= 8 ? 4 : 1), ow = 256 / zg;
    if (!room(1)) return;
    jobs.j[jobs.n++] = RedJob{part, gw, gw2, nz, XY, Ycols, mode, zg, blocks, scale};
    blocks += (XY / 4 + ow - 1) / ow;
  }
  // The five loss statistics (ppo:649-653) were a launch of one wave behind the fused heads (4.8 us on the learner stream, nothing downstream
  // reads them before the update ends): the sum over the heads' block partials rides here as one more block.
  void add_stats(const PendingStats& ps) {
    if (!room(1)) return;
    jobs.j[jobs.n++] = RedJob{ps.partials, ps.stats5, nullptr, ps.nblk, ps.N, 1, 4, 1, blocks, 1.0f};
    jobs.ent_coef = ps.ent_coef; jobs.vf_coef = ps.vf_coef;
    blocks += 1;
  }
  void launch(hipStream_t st) { if (jobs.n) hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, st, jobs); }
  // the jobs of `o` join this launch (no all-reduce waits for them: one launch instead of two)
  void absorb(const RedBatch& o) {
    if (!room(o.jobs.n)) return;
    for (int k = 0; k < o.jobs.n; ++k) { jobs.j[jobs.n] = o.jobs.j[k]; jobs.j[jobs.n].block0 += blocks; ++jobs.n; }
    blocks += o.blocks;
  }
};

// ------------------------------------------------------------------------------------------ workspace
static int dmalloc(float** p, size_t nfloats) {
  if (hipMalloc((void**)p, nfloats * sizeof(float)) != hipSuccess) { cbm_set_error("hipMalloc of %zu floats failed", nfloats); return -1; }
  return 0;
}
static int ceil_div(int a, int b) { return (a + b - 1) / b; }
static int round_up(int a, int b) { return ceil_div(a, b) * b; }

// rows of the reduction handled by one split block (multiples of BR = 32)
static const int RPS_C2 = 1280, RPS_HEADS = 128;   // (heads: 64-row slices are 3 us faster per minibatch, and their summation grouping moves the whole-update statistics from 6e-6 to 5e-5 of the oracle: not taken; RPS_C2: the im2col conv2 weight gradient of the split-bf16 mode)
// dense weight gradient: 128x256 tiles (2x4 accumulators per wave): 50 tiles x 10 reduction slices, 133 -> 124 us isolated (128x128 x 5: 133;
// 64x64 x 2: 170; 5 / 8 / 15 slices of the 128x256 tile: 143 / 148 / 160)
static const int DENSE_WGRAD_NZ = 10;
// (smaller batches: one slice per 128 frames — IMPALA's default 21 x 30 minibatch ran the 50 tiles of the 3136 x 512 gradient as 50 blocks, 116 us,
// the longest kernel of its update)
static int dense_wgrad_splits(int B) { return B >= 2048 ? DENSE_WGRAD_NZ : std::max(1, std::min(DENSE_WGRAD_NZ, ceil_div(B, 128))); }

// offsets (floats) of the per-layer partial regions inside ws.wg_part / ws.bias_part.  Built ONCE from the workspace's maxB and used for every
// batch B <= maxB: nz[i] is an upper bound of the slices any such B produces.  The frame-split counts are NOT monotone in B (they saturate at
// the block count: 256 blocks at B = 256, 129..256 for B = 257..511), so the bound is min(maxB, blocks), not splits(maxB).
struct WgRegions {
  int nz[5];            // 0 heads, 1 dense, 2 conv3, 3 conv2, 4 conv1
  size_t w[5], b[5], w_total, b_total;
  explicit WgRegions(int maxB) {
    nz[0] = ceil_div(maxB, RPS_HEADS);
    nz[1] = std::max(dense_wgrad_splits(maxB), 4);   // room for the split-bf16 mode's 4 splits
    nz[2] = conv3_wgrad_frames_splits_bound(maxB);
    nz[3] = std::max(ceil_div(maxB * 81, RPS_C2), conv2_wgrad_frames_splits_bound(maxB));
    nz[4] = conv1_wgrad_frames_splits_bound(maxB);
    const size_t wsz[5] = {512 * 32, 3136 * 512, 576 * 64, 512 * 64, 256 * 32}, bsz[5] = {32, 512, 64, 64, 32};
    size_t ow = 0, ob = 0;
    for (int i = 0; i < 5; ++i) { w[i] = ow; b[i] = ob; ow += (size_t)nz[i] * wsz[i]; ob += (size_t)nz[i] * bsz[i] + 64; }
    w_total = ow; b_total = ob;
  }
};

static int rn_ws_alloc(NatureWs& ws, int maxB, bool with_grad, int dense_ksplit_small);
int nature_ws_alloc(NatureWs& ws, int maxB, bool with_grad, int dense_ksplit_small, int kind) {
  ws.maxB = maxB; ws.with_grad = with_grad; ws.kind = kind;
  if (kind == CBM_NET_IMPALA_RESNET) return rn_ws_alloc(ws, maxB, with_grad, dense_ksplit_small);
  const size_t B = (size_t)maxB;
  if (dmalloc(&ws.act1, B * 12800) || dmalloc(&ws.act2, B * 5184) || dmalloc(&ws.act3, B * 3136) || dmalloc(&ws.hid, B * 512) ||
      dmalloc(&ws.logits, B * 32) || dmalloc(&ws.value, B)) return -1;
  ws.dense_part_ksplit = dense_ksplit_small;
  if (!with_grad) {   // actor-side workspaces: the dataflow step's arrival counters (zero = no launch yet) and its give-up word in mapped host memory
    if (hipMalloc((void**)&ws.af_cnt, 3 * B * sizeof(uint32_t)) != hipSuccess) { cbm_set_error("hipMalloc failed"); return -1; }
    hipMemset(ws.af_cnt, 0, 3 * B * sizeof(uint32_t));
    ws.af_epoch = 0;
    if (hipHostMalloc((void**)&ws.af_err_host, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&ws.af_err_dev, ws.af_err_host, 0) != hipSuccess) { cbm_set_error("hipHostMalloc failed"); return -1; }
    *ws.af_err_host = 0;
  }
  // split-K partials are only used for small batches (M <= 1024 frames: actor steps, bootstrap value)
  if (dense_ksplit_small > 1) { if (dmalloc(&ws.dense_part, (size_t)dense_ksplit_small * (B < 1024 ? B : 1024) * 512)) return -1; }
  if (with_grad) {
    if (dmalloc(&ws.dzv, B * 32) || dmalloc(&ws.dhid, B * 512) || dmalloc(&ws.dact3pad, B * 7744) || dmalloc(&ws.dact2pad, B * 7744) ||
        dmalloc(&ws.dact1, B * 12800)) return -1;
    if (hipMalloc((void**)&ws.mask1, B * 400 * 4) != hipSuccess || hipMalloc((void**)&ws.mask2, B * 81 * 2 * 4) != hipSuccess ||
        hipMalloc((void**)&ws.mask3, B * 49 * 2 * 4) != hipSuccess) { cbm_set_error("hipMalloc failed"); return -1; }
    if (hipMalloc((void**)&ws.c3_order, ((B + 127) / 128) * 81 * sizeof(int32_t)) != hipSuccess) { cbm_set_error("hipMalloc failed"); return -1; }
    hipMemset(ws.dact3pad, 0, B * 7744 * sizeof(float));
    hipMemset(ws.dact2pad, 0, B * 7744 * sizeof(float));
    hipMemset(ws.dzv, 0, B * 32 * sizeof(float));
    // split-K partials: one region per layer (WgRegions), so that the reductions can be deferred and batched
    const WgRegions rg(maxB);
    const size_t need = rg.w_total, bneed = rg.b_total;
    ws.wg_part_floats = (int64_t)need;
    ws.bias_part_floats = (int64_t)bneed;
    if (dmalloc(&ws.wg_part, need) || dmalloc(&ws.bias_part, bneed)) return -1;
  }
  return 0;
}
void nature_ws_free(NatureWs& ws) {
  float** ps[] = {&ws.act1, &ws.act2, &ws.act3, &ws.hid, &ws.logits, &ws.value, &ws.dense_part, &ws.dzv, &ws.dhid,
                  &ws.dact3pad, &ws.dact2pad, &ws.dact1, &ws.wg_part, &ws.bias_part};
  for (auto p : ps) { if (*p) hipFree(*p); *p = nullptr; }
  for (int s = 0; s < 3; ++s) {
    for (int j = 0; j < 6; ++j) { if (ws.rn_t[s][j]) hipFree(ws.rn_t[s][j]); ws.rn_t[s][j] = nullptr; }
    if (ws.rn_pidx[s]) hipFree(ws.rn_pidx[s]);
    ws.rn_pidx[s] = nullptr;
    for (int j = 0; j < 5; ++j) { if (ws.rn_m[s][j]) hipFree(ws.rn_m[s][j]); ws.rn_m[s][j] = nullptr; }
    for (int j = 0; j < 2; ++j) { if (ws.rn_tr[s][j]) hipFree(ws.rn_tr[s][j]); ws.rn_tr[s][j] = nullptr; }
  }
  for (int j = 0; j < 2; ++j) { if (ws.rn_g[j]) hipFree(ws.rn_g[j]); ws.rn_g[j] = nullptr; }
  if (ws.rn_wT) { hipFree(ws.rn_wT); ws.rn_wT = nullptr; }
  if (ws.c3_order) { hipFree(ws.c3_order); ws.c3_order = nullptr; }
  if (ws.af_cnt) { hipFree(ws.af_cnt); ws.af_cnt = nullptr; }
  if (ws.af_err_host) { hipHostFree(ws.af_err_host); ws.af_err_host = nullptr; ws.af_err_dev = nullptr; }
  for (uint32_t** m : {&ws.mask1, &ws.mask2, &ws.mask3}) { if (*m) hipFree(*m); *m = nullptr; }
  ws.c3_order_S = -1;
}

// ------------------------------------------------------------------------------------------ drivers
// every launch of the twelve GEMM kernel ids goes through prof_launch: optional HIP events around it (cbm_profile_select) and the name of the
// kernel that was ACTUALLY launched for the id (cbm_profile_kernel_name: bench.py checks it against the kernel the PMC traffic was collected on)
// demangled type of the problem functor, e.g. "DenseDgrad<IgemmTile<128, 128, 16, 2, 2, 2> >" — the same Itanium demangling a profiler prints
// inside the kernel's name, defaults included (__PRETTY_FUNCTION__ would drop defaulted template arguments)
template <class P> static const char* functor_name() {
  static const std::string s = [] {
    int status = 0;
    char* d = abi::__cxa_demangle(typeid(P).name(), nullptr, nullptr, &status);
    std::string r = status == 0 && d ? d : typeid(P).name();
    free(d);
    return r;
  }();
  return s.c_str();
}
template <class F>
static inline void prof_launch(NatureWs& ws, int kid, hipStream_t st, const char* kernel, const char* functor, F&& launch) {
  CbmProf* pf = ws.prof;
  const bool on = pf && (pf->sel == kid || pf->sel == CBM_PROF_ALL) && pf->n < CBM_PROF_MAX;
  if (on) hipEventRecord(pf->ev[2 * pf->n], st);
  launch();
  if (on) { hipEventRecord(pf->ev[2 * pf->n + 1], st); pf->kid[pf->n] = (int8_t)kid; pf->n += 1; pf->kernel[kid] = kernel; pf->functor[kid] = functor; }
}
template <class P>
static inline void plaunch(NatureWs& ws, int kid, const P& p, int nz, hipStream_t st) {
  prof_launch(ws, kid, st, "igemm_kernel", functor_name<P>(), [&] { igemm_launch(p, nz, st); });
}
template <class P>
static inline void plaunch_pf2(NatureWs& ws, int kid, const P& p, int nz, hipStream_t st) {
  prof_launch(ws, kid, st, "igemm_pf2_kernel", functor_name<P>(), [&] { igemm_pf2_launch(p, nz, st); });
}
// forward GEMMs: fp32 chain (default) or the bf16-MFMA variant when the context was created with forward_bf16
template <class P>
static inline void plaunch_fwd(NatureWs& ws, int kid, const P& p, int nz, hipStream_t st) {
  if (ws.bf16_fwd) { prof_launch(ws, kid, st, "igemm_bf16_kernel", functor_name<P>(), [&] { igemm_bf16_launch(p, nz, st); }); return; }
  if constexpr (P::DMA_OK) {
    if (p.X() >= 1024) {   // learner-size grids: tiles staged by the load unit (igemm_dma_kernel), same bits
      prof_launch(ws, kid, st, "igemm_dma_kernel", functor_name<P>(), [&] { igemm_dma_launch(p, nz, st); });
      return;
    }
  }
  plaunch(ws, kid, p, nz, st);
}
// backward GEMM launch: fp32 MFMA (default: input gradients on the two-chunk prefetch kernel igemm_pf2_kernel — dense 147 -> 143, conv3 198 -> 190,
// conv2 286 -> 268 us) or, with cbm_config.backward_split = 2 / 3, the split-bf16 kernels
template <class P>
static inline void plaunch_bwd(NatureWs& ws, int kid, const P& p, int nz, hipStream_t st) {
  // (three-term weight gradients measured slower than fp32 MFMA: with backward_split = 3 only the input gradients are split)
  if (ws.bwd_split == 0 || (P::A_RX && ws.bwd_split != 2)) {
    if constexpr (!P::A_RX) plaunch_pf2(ws, kid, p, nz, st);
    else plaunch(ws, kid, p, nz, st);
    return;
  }
  if constexpr (P::A_RX) prof_launch(ws, kid, st, "igemm_split_wgrad_kernel", functor_name<P>(), [&] { igemm_split_wgrad_launch<P, 2>(p, nz, st); });
  else prof_launch(ws, kid, st, "igemm_split_kernel", functor_name<P>(), [&] { if (ws.bwd_split == 2) igemm_split_launch<P, 2>(p, nz, st); else igemm_split_launch<P, 3>(p, nz, st); });
}
using T128x32 = IgemmTile<128, 32, 32, 4, 1>;           // heads weight gradient
using T128x32k16 = IgemmTile<128, 32, 16, 4, 1>;        // conv1 gather at 513..1023-frame batches
using T128x64 = IgemmTile<128, 64, 16, 2, 2>;           // conv3 fwd (256x64 tiles, 2x2 accumulators per wave: 167 -> 163 us, conv2 fwd 218 -> 270: not taken), conv3 dgrad
using T128x128k16 = IgemmTile<128, 128, 16, 2, 2, 2>;   // dense dgrad (143 -> 138 us), merged conv2 dgrad
using T128x256k16 = IgemmTile<128, 256, 16, 2, 2, 2>;   // dense wgrad: 2x4 accumulators per wave
using T64x64 = IgemmTile<64, 64, 32, 2, 2>;             // conv2 fwd, dense fwd
// actor-step (small batch) tiles: half the K chunk = half the LDS, so a block still finds room on CUs mostly held by learner blocks
using T64x64k16 = IgemmTile<64, 64, 16, 2, 2>;

#include "resnet_layers.inc"

int nature_forward(const NatureLayout& L, const float* P, const uint8_t* obs, const int32_t* idx, int B, int dense_ksplit,
                    NatureWs& ws, hipStream_t st, const ActorSample* sample) {
  if (L.kind == CBM_NET_IMPALA_RESNET) return resnet_forward(L, P, obs, idx, B, dense_ksplit, ws, st, sample);
  const bool small = B <= 512;
  // actor-size forward passes (no ReLU masks wanted) run on the 16x16x4 small-batch kernel (igemm.h igemm_s16_kernel), same bits
  // (forward_bf16 too: at <= 512 frames the bf16 MFMA buys nothing over these latency-bound launches, so the actor's behaviour logits stay fp32)
  if (small && !ws.mask1 && !ws.prof && !idx && sample && actor_fused_ok(L, B, dense_ksplit, ws)) {
    // actor step: conv1 .. dense as ONE dataflow launch (actor_fused_kernel), then the per-frame tail — two launches instead of five, same bits
    launch_actor_fused(L, P, obs, B, dense_ksplit, ws, st);
    hipLaunchKernelGGL(actor_tail_rows_kernel<512>, dim3(B), dim3(256), 0, st, ws.dense_part, P + L.b[3], dense_ksplit, P + L.w[4], P + L.b[4],
                       P + L.w[5], P + L.b[5], B, L.A, *sample);
    return sample->env_obs_next ? 2 : 1;
  }
  if (small && !ws.mask1 && !ws.prof) {
    Conv1Fwd<T64x64k16> p1{obs, idx, P + L.w[0], P + L.b[0], ws.act1, B * 400, nullptr};
    // conv1 on 64-row tiles (750 blocks instead of 1500): no faster alone (rollout 6.52 -> 6.53 ms), but beside the learner's CU-filling kernels half as
    // many blocks wait for a slot — pipelined step 33.87 -> 33.64 ms over six A/B pairs, IMPALA and the host-stepped path unchanged; the same 64 rows
    // for conv2 / conv3 / dense cost the rollout alone 0.1-0.4 ms and IMPALA 2-4 % (profiles/NOTES_r03_r04.md, round 4)
    igemm_s16_launch<64, 32, 32>(p1, 1, st);
    ConvFwd<T64x64k16, 4, 4, 2, 32, 64, 20, 20, 9, 9> p2{ws.act1, P + L.w[1], P + L.b[1], ws.act2, B * 81, nullptr};
    igemm_s16_launch<32, 32, 32>(p2, 1, st);   // (K chunks of 32 like conv1 / dense: 21.5 KB of LDS and 65 VGPRs per block instead of 42 KB / 129 —
    ConvFwd<T64x64k16, 3, 3, 1, 64, 64, 9, 9, 7, 7> p3{ws.act2, P + L.w[2], P + L.b[2], ws.act3, B * 49, nullptr};
    igemm_s16_launch<32, 32, 32>(p3, 1, st);   //  no slower alone, and more of these blocks find room beside the learner's: profiles/NOTES_r03_r04.md, round 4)
    if (dense_ksplit > 1) {
      DenseFwd<T64x64k16, true> pd{ws.act3, P + L.w[3], P + L.b[3], ws.dense_part, B, 3136, 512, 3136 / dense_ksplit};
      igemm_s16_launch<32, 32, 32>(pd, dense_ksplit, st);
      if (sample && L.A + 1 <= 32 && dense_ksplit <= 16) {   // split-K reduction + heads + sampling (+ the device env's step) in one launch
        hipLaunchKernelGGL(actor_tail_rows_kernel<512>, dim3(B), dim3(256), 0, st, ws.dense_part, P + L.b[3], dense_ksplit, P + L.w[4], P + L.b[4],
                           P + L.w[5], P + L.b[5], B, L.A, *sample);
        return sample->env_obs_next ? 2 : 1;
      }
      hipLaunchKernelGGL(dense_reduce_kernel, dim3(ceil_div(B * 512, 256)), dim3(256), 0, st, ws.dense_part, P + L.b[3], ws.hid, B, 512, dense_ksplit);
      if (sample && L.A + 1 <= 32) {   // more than 16 K segments: reduce first, then the 16-row tail
        hipLaunchKernelGGL(actor_tail_kernel<512>, dim3(ceil_div(B, 16)), dim3(256), 0, st, ws.hid, P + L.w[4], P + L.b[4], P + L.w[5], P + L.b[5], B, L.A,
                           *sample);
        return 1;
      }
    } else {
      DenseFwd<T64x64k16, false> pd{ws.act3, P + L.w[3], P + L.b[3], ws.hid, B, 3136, 512, 3136};
      igemm_s16_launch<32, 32, 64>(pd, 1, st);
    }
    if (!ws.skip_heads) launch_heads_fwd(ws.hid, P + L.w[4], P + L.b[4], P + L.w[5], P + L.b[5], B, L.A, 512, ws.logits, ws.value, st);
    return 0;
  }
  if (B <= 512) {   // small batches WITH masks / profiling (parity tests): conv1 through the igemm gather; larger ones on the frame-resident kernel
    Conv1Fwd<T128x32k16> p{obs, idx, P + L.w[0], P + L.b[0], ws.act1, B * 400, ws.mask1};
    plaunch(ws, K_CONV1_FWD, p, 1, st);
  } else {
    prof_launch(ws, K_CONV1_FWD, st, ws.conv1_exact_fwd ? "conv1_fwd_exact_kernel" : "conv1_fwd_planes_kernel", "",
                [&] { launch_conv1_fwd_frames(obs, idx, P + L.w[0], P + L.b[0], ws.act1, ws.mask1, B, st, ws.conv1_exact_fwd); });
  }
  if (small) {
    ConvFwd<T64x64k16, 4, 4, 2, 32, 64, 20, 20, 9, 9> p2{ws.act1, P + L.w[1], P + L.b[1], ws.act2, B * 81, ws.mask2};
    plaunch_fwd(ws, K_CONV2_FWD, p2, 1, st);
    ConvFwd<T64x64k16, 3, 3, 1, 64, 64, 9, 9, 7, 7> p3{ws.act2, P + L.w[2], P + L.b[2], ws.act3, B * 49, ws.mask3};
    plaunch_fwd(ws, K_CONV3_FWD, p3, 1, st);
  } else {   // learner-size conv2 / conv3 forward on the two-chunk prefetch kernel (bit-identical to igemm_kernel)
    ConvFwd<T64x64, 4, 4, 2, 32, 64, 20, 20, 9, 9> p2{ws.act1, P + L.w[1], P + L.b[1], ws.act2, B * 81, ws.mask2};
    static const bool c2rw = [] { const char* e = getenv("CBM_C2_RW"); return !(e && e[0] == '0'); }();   // (=0: the im2col kernel, A/B timing)
    if (!ws.bf16_fwd && c2rw) prof_launch(ws, K_CONV2_FWD, st, "conv2_fwd_regw_kernel", "", [&] { launch_conv2_fwd_regw(ws.act1, P + L.w[1], P + L.b[1], ws.act2, ws.mask2, B, st); });
    else if (!ws.bf16_fwd) plaunch_pf2(ws, K_CONV2_FWD, p2, 1, st);
    else plaunch_fwd(ws, K_CONV2_FWD, p2, 1, st);
    ConvFwd<T128x64, 3, 3, 1, 64, 64, 9, 9, 7, 7> p3{ws.act2, P + L.w[2], P + L.b[2], ws.act3, B * 49, ws.mask3};
    if (!ws.bf16_fwd) prof_launch(ws, K_CONV3_FWD, st, "conv_fwd_regw_kernel", "", [&] { launch_conv3_fwd_regw(ws.act2, P + L.w[2], P + L.b[2], ws.act3, ws.mask3, B, st); });
    else plaunch_fwd(ws, K_CONV3_FWD, p3, 1, st);
  }
  if (dense_ksplit > 1) {
    DenseFwd<T64x64k16, true> pd{ws.act3, P + L.w[3], P + L.b[3], ws.dense_part, B, 3136, 512, 3136 / dense_ksplit};
    plaunch_fwd(ws, K_DENSE_FWD, pd, dense_ksplit, st);
    hipLaunchKernelGGL(dense_reduce_kernel, dim3(ceil_div(B * 512, 256)), dim3(256), 0, st, ws.dense_part, P + L.b[3], ws.hid, B, 512,
                       dense_ksplit);
  } else {
    DenseFwd<T64x64, false> pd{ws.act3, P + L.w[3], P + L.b[3], ws.hid, B, 3136, 512, 3136};
    plaunch_fwd(ws, K_DENSE_FWD, pd, 1, st);
  }
  if (!ws.skip_heads) launch_heads_fwd(ws.hid, P + L.w[4], P + L.b[4], P + L.w[5], P + L.b[5], B, L.A, 512, ws.logits, ws.value, st);
  return false;
}

// Tile order of the position-major conv3 dgrad.  igemm_kernel hands XCD k the contiguous run of x-tiles [start_k, start_k + len_k)
// (block b runs on XCD b % 8).  Within a run the tiles go frame-tile by frame-tile (81 pixels that read the same BX frames, L2-friendly);
// the last two frame-tiles' worth of every run is sorted by tap count, heaviest first, so that the blocks still running when the
// kernel drains are the 1- and 2-tap border pixels (4-8 K-chunks) rather than 9-tap interior ones (36): a lone block is latency-bound,
// and the drain used to cost as much as the skipped taps saved.
static void conv3_order_build(NatureWs& ws, int S, int BX, hipStream_t st) {
  if (ws.c3_order_S == S) return;
  const int nft = (S + BX - 1) / BX, nb = nft * 81;
  auto taps = [](int p) {
    const int ih = p / 9, iw = p % 9;
    return (std::min(2, 8 - ih) - std::max(0, 2 - ih) + 1) * (std::min(2, 8 - iw) - std::max(0, 2 - iw) + 1);
  };
  std::vector<int32_t> h(nb);
  for (int i = 0; i < nb; ++i) h[i] = i;
  const int q = nb >> 3, r = nb & 7;
  for (int k = 0; k < 8; ++k) {
    const int lo = k < r ? k * (q + 1) : r * (q + 1) + (k - r) * q, len = k < r ? q + 1 : q;
    const int tail = std::min(len, 162);
    std::stable_sort(h.begin() + lo + len - tail, h.begin() + lo + len, [&](int a, int b) { return taps(a % 81) > taps(b % 81); });
  }
  hipMemcpyAsync(ws.c3_order, h.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st);
  hipStreamSynchronize(st);   // h goes out of scope; happens once per batch size
  ws.c3_order_S = S;
}

// dense input gradient on 128x64 tiles (1470 blocks, three dispatch waves) instead of 128x128 (750 blocks, one and a half): 7 us slower alone (125 vs
// 118), 13 us faster beside the rollout (156 -> 143), step -0.03 ... -0.19 ms in five A/B pairs — shorter blocks in more waves suffer less from the CUs the
// rollout slows down (tools/block_trace.py)
// Round 6, K chunks of 32 for the dense and conv3 input gradients: timing builds of igemm_pf2_kernel (-DPF2_ABL, profiles/r06_pf2_ablation.txt) put 14-17 % of
// these kernels into their global loads — not the latency (a true two-chunk prefetch changed nothing) but the request count: with 16-wide chunks a row
// contributes 64 bytes per chunk, half a cache line, twice.  dense dgrad 125 -> 116 us (128x64x32, four waves per SIMD), conv3 dgrad 143.6 -> 141.2
// (two per SIMD); the merged conv2 dgrad is SLOWER on 32-wide chunks (222 -> 228: 66 KB of LDS per block) and keeps 16.
using T128x64k32 = IgemmTile<128, 64, 32, 2, 2, 2>;
using T128x64k32w4 = IgemmTile<128, 64, 32, 2, 2, 4>;
// (other shapes tried through -DCBM_DD_TILE / -DCBM_C3D_TILE / -DCBM_C2D_TILE and not kept: 128x128x32, 64x128x16, 64x64x16 and x32, 128x64x16 at three and four
// waves per SIMD, a 4x1 wave grid — results in profiles/r06_isa_fixes_ab.txt)
#ifndef CBM_DD_TILE
#define CBM_DD_TILE T128x64k32w4
#endif
#ifndef CBM_C3D_TILE
#define CBM_C3D_TILE T128x64k32
#endif
#ifndef CBM_C2D_TILE
#define CBM_C2D_TILE T128x64     // (round 6: two column tiles of 64 per pixel tile, four waves per SIMD: 224 -> 214 us against the 128x128 tile; 128x64x32 218-231, 64x128 228)
#endif
void nature_backward(const NatureLayout& L, const float* P, const uint8_t* obs, const int32_t* idx, int B, NatureWs& ws, float* grads,
                     hipStream_t st) {
  if (L.kind == CBM_NET_IMPALA_RESNET) { resnet_backward(L, P, obs, idx, B, ws, grads, st); return; }
  const int A = L.A;
  const WgRegions rg(ws.maxB);   // the layout the workspace was allocated with; every slice count below is checked against it
  auto fits = [&](int layer, int nz) {
    if (nz <= rg.nz[layer]) return true;
    cbm_launch_fail("wgrad partial region %d: %d slices > %d allocated (workspace sized for %d frames, batch %d)", layer, nz, rg.nz[layer], ws.maxB, B);
    return false;
  };
  float* const wp = ws.wg_part;
  float* const bp = ws.bias_part;
  RedBatch tail_red(A), conv_red(A);
  // heads: dgrad (VALU) and wgrad (MFMA, Y = A+1 padded to 32)
  if (!ws.skip_heads) launch_heads_dgrad(ws.dzv, P + L.w[4], P + L.w[5], ws.hid, B, A, 512, ws.dhid, st);
  {
    const int nz = ceil_div(B, RPS_HEADS);
    if (!fits(0, nz)) return;
    MatWgrad<T128x32> p{ws.hid, ws.dzv, wp + rg.w[0], bp + rg.b[0], B, 512, 32, 32, RPS_HEADS};
    plaunch(ws, K_HEADS_WGRAD, p, nz, st);   // (fp32 also in split mode: 9 vs 15 us)
    tail_red.add(wp + rg.w[0], nz, 512 * 32, 32, 2, grads + L.w[4], grads + L.w[5]);
    tail_red.add(bp + rg.b[0], nz, 32, 32, 3, grads + L.b[4], grads + L.b[5]);
  }
  // dense: dgrad -> dact3pad, wgrad
  {
    DenseDgrad<CBM_DD_TILE> pd{ws.dhid, P + L.w[3], ws.act3, ws.dact3pad, B, ws.mask3};
    plaunch_bwd(ws, K_DENSE_DGRAD, pd, 1, st);
    // split-bf16 mode: the staging of a tile is the bottleneck, so it wants the bigger 128x64 tile (and more splits to fill the chip)
    const int nz = ws.bwd_split == 2 ? (B >= 2048 ? 4 : 1) : (dense_wgrad_dma_slices(B) ? dense_wgrad_dma_slices(B) : dense_wgrad_splits(B));
    const int rps = round_up(ceil_div(B, nz), 32);
    if (!fits(1, nz)) return;
    if (ws.bwd_split == 2) {
      MatWgrad<T128x64> pw{ws.act3, ws.dhid, wp + rg.w[1], bp + rg.b[1], B, 3136, 512, 512, rps};
      plaunch_bwd(ws, K_DENSE_WGRAD, pw, nz, st);
    } else if (dense_wgrad_dma_slices(B)) {   // learner minibatches: both operands straight from the load unit (dense_wgrad.hip)
      prof_launch(ws, K_DENSE_WGRAD, st, "dense_wgrad_dma_kernel", "", [&] { launch_dense_wgrad_dma(ws.act3, ws.dhid, wp + rg.w[1], bp + rg.b[1], B, 3136, 512, nz, st); });
    } else {
      MatWgrad<T128x256k16> pw{ws.act3, ws.dhid, wp + rg.w[1], bp + rg.b[1], B, 3136, 512, 512, rps};
      plaunch(ws, K_DENSE_WGRAD, pw, nz, st);
    }
    tail_red.add(wp + rg.w[1], nz, 3136 * 512, 512, 0, grads + L.w[3], nullptr);
    tail_red.add(bp + rg.b[1], nz, 512, 512, 0, grads + L.b[3], nullptr);
    if (ws.tail_ev) tail_red.launch(st);   // (without a communicator nobody waits for the tail: its reductions ride in the launch at the end)
  }
  if (ws.tail_ev) hipEventRecord(ws.tail_ev, st);   // 95 % of the flat gradient is final here: its all-reduce can overlap the conv backward
  // conv3: dgrad -> dact2pad (position-major with tap skipping), wgrad
  {
    conv3_order_build(ws, B, CBM_C3D_TILE::BX, st);
    Conv3DgradPos<CBM_C3D_TILE> pd{ws.dact3pad, P + L.w[2], ws.act2, ws.dact2pad, B, ws.c3_order, ws.mask2};
    plaunch_bwd(ws, K_CONV3_DGRAD, pd, 1, st);
    // frame-resident kernel (wgrad_frames.hip): the whole 576x64 gradient in the block's accumulators, act2 / dY frames copied once into LDS
    // (fp32 MFMA also in split mode: the split weight-gradient kernel measured slower than the im2col fp32 one already)
    const int nz = conv3_wgrad_frames_splits(B);
    if (!fits(2, nz)) return;
    prof_launch(ws, K_CONV3_WGRAD, st, "conv3_wgrad_frames_kernel", "", [&] { launch_conv3_wgrad_frames(ws.act2, ws.dact3pad, wp + rg.w[2], bp + rg.b[2], B, st); });
    conv_red.add(wp + rg.w[2], nz, 576 * 64, 64, 0, grads + L.w[2], nullptr);
    conv_red.add(bp + rg.b[2], nz, 64, 64, 0, grads + L.b[2], nullptr);
  }
  // conv2: dgrad -> dact1 (merged position-major form, N = 128, one dY tile for the four classes: 268 -> 246 us), wgrad
  {
    Conv2DgradMergedPos<CBM_C2D_TILE> pd{ws.dact2pad, P + L.w[1], ws.dact1, B, ws.mask1};
    plaunch_bwd(ws, K_CONV2_DGRAD, pd, 1, st);
    if (ws.bwd_split != 2) {   // frame-resident kernel
      const int nz = conv2_wgrad_frames_splits(B);
      if (!fits(3, nz)) return;
      prof_launch(ws, K_CONV2_WGRAD, st, "conv2_wgrad_frames_kernel", "", [&] { launch_conv2_wgrad_frames(ws.act1, ws.dact2pad, wp + rg.w[3], bp + rg.b[3], B, st); });
      conv_red.add(wp + rg.w[3], nz, 512 * 64, 64, 0, grads + L.w[1], nullptr);
      conv_red.add(bp + rg.b[3], nz, 64, 64, 0, grads + L.b[1], nullptr);
    } else {                   // split-bf16: im2col GEMM; the split kernel wants the bigger tile (staging-bound)
      const int M = B * 81, nz = ceil_div(M, RPS_C2);
      if (!fits(3, nz)) return;
      ConvWgrad<T128x64, 4, 4, 2, 32, 64, 20, 20, 9, 9, 1> pw{ws.act1, ws.dact2pad, wp + rg.w[3], bp + rg.b[3], M, RPS_C2};
      plaunch_bwd(ws, K_CONV2_WGRAD, pw, nz, st);
      conv_red.add(wp + rg.w[3], nz, 512 * 64, 64, 0, grads + L.w[1], nullptr);
      conv_red.add(bp + rg.b[3], nz, 64, 64, 0, grads + L.b[1], nullptr);
    }
  }
  // conv1: wgrad only (frames need no gradient); frame-resident kernel, pixels as integers, 1/255 in the reduce
  {
    const int nz = conv1_wgrad_frames_splits(B);
    if (!fits(4, nz)) return;
    const bool c1x = ws.conv1_exact_wgrad && B > 512 && ws.bwd_split != 2;
    prof_launch(ws, K_CONV1_WGRAD, st, c1x ? "conv1_wgrad_exact_kernel" : ws.bwd_split == 2 ? "conv1_wgrad_frames_split_kernel" : "conv1_wgrad_frames_kernel", "",
                [&] { launch_conv1_wgrad_frames(obs, idx, ws.dact1, wp + rg.w[4], bp + rg.b[4], B, st, ws.bwd_split == 2, c1x); });
    conv_red.add(wp + rg.w[4], nz, 256 * 32, 32, 1, grads + L.w[0], nullptr, 1.0f / 255.0f);
    conv_red.add(bp + rg.b[4], nz, 32, 32, 0, grads + L.b[0], nullptr);
    if (!ws.tail_ev) conv_red.absorb(tail_red);
    if (ws.pending_stats.partials) { conv_red.add_stats(ws.pending_stats); ws.pending_stats = PendingStats{}; }
    conv_red.launch(st);
  }
}
