// conv_regw.hip — learner-size conv3 forward (3x3 stride 1, 64 -> 64 channels, 9x9 -> 7x7; naturecnn:160-166) and conv2 forward (further down) with the WEIGHTS IN REGISTERS.
//
// The im2col GEMM kernels (igemm.h) stage both operands through LDS tile by tile: per 16-wide K chunk a gather, a register -> LDS store, a barrier.
// Here nothing but the activations ever goes through LDS and nothing but `ds_read_b32` + MFMA is issued inside a step:
//   * a block is 8 waves; wave (cq, tg) keeps W[0:576][16*cq : 16*cq+16] — its 16 output channels, ALL of K — in 144 VGPRs as the B fragments of
//     v_mfma_f32_16x16x4_f32 (lane (g4, r16) holds W[4s + g4][16 cq + r16] for step s).  147 KB of weights live in the CU's register file, read once.
//   * the block owns a contiguous run of frames and walks their 49-position outputs as ONE stream of 16-position tiles (a tile may straddle two
//     frames: every lane carries its own pixel address), 4 tiles = 64 positions per step; waves tg = 0 / 1 take tiles {0,1} / {2,3} of the step (one tile each in a short last step) with
//     two independent accumulators, the four cq waves of a tile read the same A fragments.
//   * input frames sit in a ring of 7 LDS slots as NHWC pixels of 64 floats at a pitch of 66: lane (g4, r16) reads pixel(r16) * 66 + 4c + g4.  A
//     ds_read_b32 is served in two groups of 32 lanes over 32 banks (MI355X_MICROARCH, LDS): lanes 0-31 are g4 = 0 / 1 x 16 positions; pixel pitch = 2,
//     input-row pitch = 2 * 7 and slot pitch = 2 * 49 (mod 32) put ANY 16 consecutive positions of the stream on banks 2 p + g4 — conflict-free
//     (pixel pitch 68: 2-way, PMC conflict share 67 %, 124.5 us; 66 with plain row / slot pitches: tiles spanning output rows collide, 50 %, 122.2 us;
//     now 120.2 us).
//   *  A frame is copied by 81 `global_load_lds_dword` (one pixel = 256 B each), issued a
//     whole step (~10 us) before its first use; one raw s_barrier per step, two thirds into it (see SYNC_TAP); the step's results are stored at the
//     top of the NEXT step, so the vmcnt(0) in front of the barrier never waits for a store or a copy issued less than half a step ago.
//   * a step's K loop is 9 taps x 16 channel quads = 144 x (2 ds_read_b32 with immediate offsets + 2 MFMA): k = (kh, kw, ci) ascending in one accumulator, four k
//     per instruction in ascending order — the same fmaf chain as igemm_kernel / igemm_s16_kernel, so the forward stays bit-exact against the oracle.
// Epilogue = ConvFwd::store_flag's: relu(acc + bias) and the ReLU bit mask (one uint32 per position and 32 channels; a wave owns 16 of them and writes its
// half as a uint16).
#include "cbm_internal.h"
#include <type_traits>

typedef float f32x4_rw __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void rw_glds4(const float* g_lane, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane, (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

namespace {
struct C3G {
  static constexpr int KH = 3, KW = 3, CI = 64, CO = 64, IH = 9, IW = 9, OH = 7, OW = 7;
  static constexpr int PIX = IH * IW, NPOS = OH * OW, PP = CI + 2, NS = 7, NSTEP = KH * KW * CI / 4;
  static constexpr int SYNC_TAP = 6;                          // the step's barrier sits in front of this tap
  static constexpr int NTG = 2, NW = 4 * NTG, DPW = (PIX + NW - 1) / NW;     // tile groups (waves per SIMD), waves, copies per wave and frame
  static constexpr int SP = 32 * NTG;                          // positions per step
  // bank of a fragment read = (address / 4) mod 32 within a group of 32 lanes (16 positions x k = 4s + {0, 1}): pixel pitch = 2 (mod 32) puts the positions
  // of one output row on banks 2 ox + k; the input-ROW pitch continues that sequence into the next output row (OW positions later: = 2 OW mod 32) and the
  // slot pitch into the next frame (= 2 NPOS mod 32), so that any 16 consecutive positions of the stream sit on 32 different banks
  static constexpr int RP0 = IW * PP, RP = RP0 + ((2 * OW - RP0) % 32 + 32) % 32;
  static constexpr int SLOT0 = IH * RP, SLOT = SLOT0 + ((2 * NPOS - SLOT0) % 32 + 32) % 32;
  static_assert(RP % 32 == (2 * OW) % 32 && SLOT % 32 == (2 * NPOS) % 32 && PP % 32 == 2, "bank sequence");
  static constexpr int LDS_BYTES = NS * SLOT * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "frame ring exceeds the LDS");
};
}  // namespace

template <class G>
__global__ __launch_bounds__(64 * G::NW) void conv_fwd_regw_kernel(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                                                            float* __restrict__ out, uint32_t* __restrict__ mask, int B, int fpb) {
  extern __shared__ __attribute__((aligned(16))) float rw_lds[];
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cq = wave & 3, tg = wave >> 2;
  const int f0 = blockIdx.x * fpb, nf = min(fpb, B - f0);
  if (nf <= 0) return;
  const int P = nf * G::NPOS, nsteps = (P + G::SP - 1) / G::SP;
  const float* src0 = in + (size_t)f0 * (G::PIX * G::CI);

  int issued = 0;
  auto issue_upto = [&](int lim) __attribute__((always_inline)) {
    lim = min(lim, nf);
    for (; issued < lim; ++issued) {
      const float* src = src0 + (size_t)issued * (G::PIX * G::CI) + lane;
      float* dst = rw_lds + (issued % G::NS) * G::SLOT;
#pragma unroll
      for (int i = 0; i < G::DPW; ++i) {
        const int p = min(wave * G::DPW + i, G::PIX - 1);      // (the last wave repeats pixel 80: same bytes to the same place)
        rw_glds4(src + p * G::CI, dst + (p / G::IW) * G::RP + (p % G::IW) * G::PP);
      }
    }
  };
  // Only the frames the first step and a third need (reads before the second barrier touch positions < 2 SP) are requested up front; the rest of the ring
  // is filled from the first barrier on (all 256 blocks start together: seven frames each are 37 MB wanted at once).
  issue_upto((2 * G::SP - 1) / G::NPOS + 1);

  float w[G::NSTEP];
#pragma unroll
  for (int s = 0; s < G::NSTEP; ++s) w[s] = W[(size_t)(4 * s + g4) * G::CO + 16 * cq + r16];
  const float bv = bias[16 * cq + r16];
  uint16_t* mask16 = reinterpret_cast<uint16_t*>(mask);

  // A step = 64 positions = 4 tiles, two per wave (tg takes tiles 2 tg, 2 tg + 1).  When the LAST step has 32 positions or fewer it runs as a half step:
  // one tile per wave (tile tg), half the instructions — at 3840 frames a block owns 735 positions = 11.5 steps, and a full twelfth step was 4 % of the kernel.
  const bool half_last = P - G::SP * (nsteps - 1) <= G::SP / 2;
  float o[2][4];
  uint32_t mh[2] = {0u, 0u};
  auto flush = [&](int t, bool half) __attribute__((always_inline)) {      // results of step t
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (half && j) break;
      const int q0 = G::SP * t + (half ? tg : 2 * tg + j) * 16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = q0 + 4 * g4 + e;
        if (q < P) out[((size_t)f0 * G::NPOS + q) * G::CO + 16 * cq + r16] = o[j][e];
      }
      if (mask16 && r16 < 4) {
        const int q = q0 + 4 * g4 + r16;
        if (q < P) mask16[((size_t)f0 * G::NPOS + q) * (G::CO / 16) + cq] = (uint16_t)mh[j];
      }
    }
  };

  constexpr int QPT = G::CI / 4, NTAP = G::KH * G::KW;          // channel quads per tap
  auto step = [&](auto ntc, int t) __attribute__((always_inline)) {
    constexpr int NT = decltype(ntc)::value;
    int base[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int q = min(G::SP * t + (NT == 1 ? tg : 2 * tg + j) * 16 + r16, P - 1);
      const int fr = q / G::NPOS, p = q - fr * G::NPOS, oy = p / G::OW, ox = p - oy * G::OW;
      base[j] = (fr % G::NS) * G::SLOT + oy * G::RP + ox * G::PP + g4;
    }
    f32x4_rw acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4_rw{0.f, 0.f, 0.f, 0.f};
    // A fragments of a whole tap (QPT k-steps x NT tiles) are read while the previous tap is multiplied: the reads are a tap (~1000 cycles) ahead
    float a[2][NT][QPT];
    auto load_part = [&](int tap, int c0, float (&dst)[NT][QPT]) __attribute__((always_inline)) {   // quads c0 .. c0+3 of a tap
      const int kh = tap / G::KW, kw = tap - kh * G::KW, off = kh * G::RP + kw * G::PP;
#pragma unroll
      for (int c = c0; c < c0 + 4; ++c)
#pragma unroll
        for (int j = 0; j < NT; ++j) dst[j][c] = rw_lds[base[j] + off + 4 * c];
    };
#pragma unroll
    for (int c0 = 0; c0 < QPT; c0 += 4) load_part(0, c0, a[0]);
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      if (tap == G::SYNC_TAP) {
        // Two thirds into the step: every copy issued at this point of the previous step has had a whole step to land (the wait is free), and past
        // the barrier every wave has left step t-1, so the frames below lo(t) can be overwritten.  What is issued here is first read after the NEXT
        // barrier.  Keeping the barrier away from the step boundary lets the two waves of a SIMD drift apart: one wave's stores / address arithmetic /
        // first LDS round trip run under the other's MFMAs.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        issue_upto((G::SP * t) / G::NPOS + G::NS);
      }
#pragma unroll
      for (int c0 = 0; c0 < QPT; c0 += 4) {
        __builtin_amdgcn_sched_barrier(0);                       // (the scheduler otherwise sinks every read to just before its use)
        if (tap + 1 < NTAP) load_part(tap + 1, c0, a[(tap + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = c0; c < c0 + 4; ++c)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap & 1][j][c], w[tap * QPT + c], acc[j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      uint64_t b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pre = acc[j][e] + bv;
        const float v = pre > 0.0f ? pre : 0.0f;
        o[j][e] = v;
        b[e] = __ballot(v > 0.0f);
      }
      const uint64_t mine = r16 == 0 ? b[0] : (r16 == 1 ? b[1] : (r16 == 2 ? b[2] : b[3]));   // row 4 g4 + r16 lives in ballot r16, bits [16 g4, +16)
      mh[j] = (uint32_t)(mine >> (16 * g4)) & 0xffffu;
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");                       // the first NS frames are in LDS
  const int nfull = half_last ? nsteps - 1 : nsteps;
  for (int t = 0; t < nfull; ++t) {
    if (t > 0) flush(t - 1, false);
    step(std::integral_constant<int, 2>{}, t);
  }
  if (half_last) {
    if (nfull > 0) flush(nfull - 1, false);
    step(std::integral_constant<int, 1>{}, nfull);
  }
  flush(nsteps - 1, half_last);
}

// ------------------------------------------------------------------------------------------------ conv2 forward (4x4 stride 2, 32 -> 64, 20x20 -> 9x9; naturecnn:152-158)
// Same recipe; what differs: K = 512 -> 128 weight registers per wave; a frame is 51 KB, so the ring holds input ROWS (20 per frame, 10 copy granules of 2
// pixels = 64 floats each), three mirror rows behind it keep a lane's four tap rows contiguous; stride 2 means consecutive output positions are one GRANULE
// apart, so a granule pitch of 66 = 2 (mod 32) puts them on banks 2 p + k, and two input rows (one output row) = 2 * 9 (mod 32) continues the sequence.
namespace {
struct C2G {
  static constexpr int KH = 4, KW = 4, ST = 2, CI = 32, CO = 64, IH = 20, IW = 20, OH = 9, OW = 9;
  static constexpr int NPOS = OH * OW, NSTEP = KH * KW * CI / 4, QPT = CI / 4;
  static constexpr int NGR = IW * CI / 64, GF = 66;                       // copy granules per input row, granule pitch
  static constexpr int RP0 = NGR * GF, RP = RP0 + ((OW - RP0) % 16 + 16) % 16;   // 2 RP = 2 OW (mod 32)
  static_assert((2 * RP) % 32 == (2 * OW) % 32 && GF % 32 == 2, "bank sequence");
  static constexpr int NTG = 2, NW = 4 * NTG, SP = 32 * NTG;
  static constexpr int NR = 54, NRP = NR + 3, LDS_BYTES = NRP * RP * 4;   // (ring_ok: three steps of positions touch up to 54 input rows)
  static constexpr int SYNC_TAP = 11;                                     // of 16
  static_assert(LDS_BYTES <= 160 * 1024, "row ring exceeds the LDS");
  __host__ __device__ static int row_of(int q) { const int f = q / NPOS, p = q - f * NPOS; return f * IH + ST * (p / OW); }   // first input row of position q
  __host__ __device__ static int need_lo(int t, int P) { const int q = SP * t < P - 1 ? SP * t : P - 1; return row_of(q); }
  __host__ __device__ static int need_hi(int t, int P) { const int q = (SP * (t + 1) < P ? SP * (t + 1) : P) - 1; return row_of(q) + KH - 1; }
  static bool ring_ok(int nf) {
    const int P = nf * NPOS, ns = (P + SP - 1) / SP;
    if (need_hi(ns > 1 ? 1 : 0, P) + 1 > NR) return false;
    for (int t = 1; t + 1 < ns; ++t) if (need_hi(t + 1, P) >= need_lo(t - 1, P) + NR) return false;
    return true;
  }
};
}  // namespace

__global__ __launch_bounds__(512) void conv2_fwd_regw_kernel(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                                                             float* __restrict__ out, uint32_t* __restrict__ mask, int B, int fpb) {
  using G = C2G;
  extern __shared__ __attribute__((aligned(16))) float rw_lds[];
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cq = wave & 3, tg = wave >> 2;
  const int f0 = blockIdx.x * fpb, nf = min(fpb, B - f0);
  if (nf <= 0) return;
  const int P = nf * G::NPOS, nsteps = (P + G::SP - 1) / G::SP, NV = nf * G::IH;
  const float* src0 = in + (size_t)f0 * (G::IH * G::IW * G::CI) + lane;

  int issued = 0;                                                        // input rows [0, issued) of the block's frames have been requested
  auto issue_row = [&](int v) __attribute__((always_inline)) {          // one wave copies a row: 10 granules at constant offsets
    const float* src = src0 + (size_t)v * (G::IW * G::CI);
    const int pr = v % G::NR;
    float* dst = rw_lds + pr * G::RP;
#pragma unroll
    for (int g = 0; g < G::NGR; ++g) rw_glds4(src + g * 64, dst + g * G::GF);
    if (pr < 3) {
      float* dm = dst + G::NR * G::RP;
#pragma unroll
      for (int g = 0; g < G::NGR; ++g) rw_glds4(src + g * 64, dm + g * G::GF);
    }
  };
  auto issue_upto = [&](int lim) __attribute__((always_inline)) {
    lim = min(lim, NV);
    for (int v = issued + wave; v < lim; v += G::NW) issue_row(v);
    issued = max(issued, lim);
  };
  issue_upto(G::need_hi(nsteps > 1 ? 1 : 0, P) + 1);

  float w[G::NSTEP];
#pragma unroll
  for (int s = 0; s < G::NSTEP; ++s) w[s] = W[(size_t)(4 * s + g4) * G::CO + 16 * cq + r16];
  const float bv = bias[16 * cq + r16];
  uint16_t* mask16 = reinterpret_cast<uint16_t*>(mask);
  const size_t Q0 = (size_t)f0 * G::NPOS;

  auto sync_and_issue = [&](int t) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    issue_upto(G::need_lo(t, P) + G::NR);
  };

  // results of a step are kept in registers and stored at the top of the next one (see conv_fwd_regw_kernel)
  float o[2][4];
  uint32_t mh[2] = {0u, 0u};
  auto flush = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q0 = G::SP * t + (2 * tg + j) * 16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = q0 + 4 * g4 + e;
        if (q < P) out[(Q0 + q) * G::CO + 16 * cq + r16] = o[j][e];
      }
      if (mask16 && r16 < 4) {
        const int q = q0 + 4 * g4 + r16;
        if (q < P) mask16[(Q0 + q) * (G::CO / 16) + cq] = (uint16_t)mh[j];
      }
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  for (int t = 0; t < nsteps; ++t) {
    if (t > 0) flush(t - 1);
    int base[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t q = min((uint32_t)(G::SP * t + (2 * tg + j) * 16 + r16), (uint32_t)(P - 1));
      const uint32_t f = q / G::NPOS, p = q - f * G::NPOS, oy = p / G::OW, ox = p - oy * G::OW;
      const uint32_t sr = (f * G::IH + G::ST * oy) % G::NR;              // ring slot of the window's first row
      base[j] = (int)(sr * G::RP + ox * G::GF + g4);                    // pixel 2 ox = granule ox
    }
    f32x4_rw acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[j] = f32x4_rw{0.f, 0.f, 0.f, 0.f};
    constexpr int QC = 4, CPT = G::QPT / QC, NCH = G::KH * G::KW * CPT;    // chunks of 4 channel quads: half a tap
    float a[2][2][QC];
    auto load_chunk = [&](int u, float (&dst)[2][QC]) __attribute__((always_inline)) {
      const int tap = u / CPT, h = u - tap * CPT, kh = tap / G::KW, kw = tap - kh * G::KW;
      const int off = kh * G::RP + (kw >> 1) * G::GF + (kw & 1) * G::CI + 4 * (h * QC);
#pragma unroll
      for (int c = 0; c < QC; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) dst[j][c] = rw_lds[base[j] + off + 4 * c];
    };
    load_chunk(0, a[0]);
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      if (u == G::SYNC_TAP * CPT) {
        __builtin_amdgcn_sched_barrier(0);
        sync_and_issue(t);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (u + 1 < NCH) load_chunk(u + 1, a[(u + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < QC; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 1][j][c], w[u * QC + c], acc[j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint64_t b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pre = acc[j][e] + bv;
        const float v = pre > 0.0f ? pre : 0.0f;
        o[j][e] = v;
        b[e] = __ballot(v > 0.0f);
      }
      const uint64_t mine = r16 == 0 ? b[0] : (r16 == 1 ? b[1] : (r16 == 2 ? b[2] : b[3]));
      mh[j] = (uint32_t)(mine >> (16 * g4)) & 0xffffu;
    }
  }
  flush(nsteps - 1);
}

void launch_conv2_fwd_regw(const float* in, const float* W, const float* bias, float* out, uint32_t* mask, int B, hipStream_t st) {
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)conv2_fwd_regw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C2G::LDS_BYTES); attr = true; }
  const int fpb = (B + 255) / 256, blocks = (B + fpb - 1) / fpb;
  static int checked = 0;
  if (checked != fpb) {
    if (!C2G::ring_ok(fpb)) { cbm_launch_fail("conv2_fwd_regw_kernel: row ring of %d slots too small at %d frames per block", C2G::NR, fpb); return; }
    checked = fpb;
  }
  hipLaunchKernelGGL(conv2_fwd_regw_kernel, dim3(blocks), dim3(512), C2G::LDS_BYTES, st, in, W, bias, out, mask, B, fpb);
}

void launch_conv3_fwd_regw(const float* in, const float* W, const float* bias, float* out, uint32_t* mask, int B, hipStream_t st) {
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)conv_fwd_regw_kernel<C3G>, hipFuncAttributeMaxDynamicSharedMemorySize, C3G::LDS_BYTES); attr = true; }
  const int fpb = (B + 255) / 256, blocks = (B + fpb - 1) / fpb;
  hipLaunchKernelGGL(conv_fwd_regw_kernel<C3G>, dim3(blocks), dim3(64 * C3G::NW), C3G::LDS_BYTES, st, in, W, bias, out, mask, B, fpb);
}
