// rnconv_rw.h — learner-size 3x3 SAME convolutions of the IMPALA-ResNet torso (ppo:149-189) with the WEIGHTS IN REGISTERS and the activations in a
// load-unit-fed NHWC row ring.  Drop-in for rn_conv_kernel (rnconv.h) at large batches: same arguments, same epilogues (EPI 0-6), same k-ascending
// v_mfma_f32_16x16x4_f32 chain k = (kh, kw, ci) -> the same bits.
//
// rn_conv_kernel stages every strip through registers into channel planes (global load -> relu / select -> four scattered ds_write per float4), reads
// A and B fragments from LDS and decodes an output index per stored element: 4.3 VALU instructions per MFMA on the 16-channel layers, two block
// barriers per strip with the matrix pipe idle in between.  Here:
//   * every wave keeps ALL weights of the layer as B fragments in VGPRs (9 CI / 4 steps x CO / 16 column blocks: 36 ... 144 registers);
//   * input rows live in LDS as NHWC pixels, copied by global_load_lds_dword in granules of 64 floats (4 pixels at CI = 16, 2 at CI = 32) at a
//     pitch of 66; a wave's tiles interleave over its positions so that a fragment read touches one pixel per granule (bank rule: see the step
//     body).  A row has RPX >= H + 1 pixel slots; the slots past H stay zero (the copy masks those lanes) and serve as the right / left halo; the
//     halo ROWS between frames are copied from a zero page, so SAME padding costs no instruction in the loop;
//   * rows form a ring of NR slots over the block's VIRTUAL rows (H + 1 per frame: a zero row, then the frame's rows), two mirror rows at the end keep
//     the three rows of a tap window contiguous; a block walks the H*H positions of its frames as one stream of 16-position tiles, TSP positions per
//     step; one raw s_barrier per step (in front of tap SYNC_TAP), rows are requested a whole step before their first use;
//   * the granule pad breaks the uniform pixel stride, so a lane carries three bases per tile (one per kw); kh, the channel quad, the four output
//     elements of a lane and the column block are immediates on 32-bit element offsets;
//   * the relu masks of the input gradients travel as one bit per element (MASKIN / mask_out below).
// The relu in front of a residual block's first conv is a v_max on the fragment (the load unit cannot apply it).
// Measured per 3840-frame minibatch against rn_conv_kernel (us): 32->32 @ 11x11 88-98 vs 127-145; 32->32 @ 21x21 267-312 vs 285-368; 16->32 @ 42x42 557
// vs 737; 32->16 @ 42x42 (input gradient) 548 vs 612; 16->16 @ 42x42 325-394 vs 370-413.  Timing builds of an earlier version: no row copies 8 %
// faster, fragments from registers instead of LDS 3-12 %, no residual / mask loads 6-7 %, no stores 1-2 %, all four 276 / 240 us (16->16 @ 42 /
// 32->32 @ 21) = the MFMA + address arithmetic + barrier skeleton.
#pragma once
#include <type_traits>
#ifndef RW_LDS_RELU   // 1: relu of a PRE_RELU kernel's input applied once per landed row in LDS (relu_rows below) instead of on every fragment.  Measured,
#define RW_LDS_RELU 0 // same bits: 16 -> 16 @ 42x42 397 -> 388 us, 32 -> 32 @ 21x21 311 -> 314: the pass sits in front of the step's barrier and costs most of
#endif                // the 61 / 25 us the 144 v_max per step cost (profiles/r06_resnet_sinks.txt); off
#ifndef RW_ABL   // timing builds only (tools/variants.sh): 1 no per-step address decode, 2 no relu on the fragments
#define RW_ABL 0
#endif

__device__ float rn_rw_zero_row[42 * 32];

template <int CI_, int CO_, int H_, int NT_, int NR_>
struct RnRwGeom {
  static constexpr int CI = CI_, CO = CO_, H = H_, NT = NT_, NR = NR_;
  static constexpr int GP = 64 / CI, GF = 66, RPX = (H + 1 + GP - 1) / GP * GP, NG = RPX / GP;
  static constexpr int NCO = CO / 16, QPT = CI / 4, NSTEP = 9 * QPT;
  static constexpr int NW = 8, TS = NW * NT, TSP = 16 * TS;     // tiles / positions per step
  static constexpr int NRP = NR + 2;                             // + two mirror rows
  static constexpr int LDS_FLOATS = (1 + NRP * NG) * GF;         // (granule 0 = the left halo of physical row 0)
  static constexpr int ROWF = H * CI, FP = H * H;
  static constexpr int SYNC_TAP = 6;
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "row ring exceeds the 160 KB LDS");
  static_assert(ROWF <= 42 * 32, "zero page too small");
  __host__ __device__ static int vc_of(int Gr) { const int f = Gr / H; return f * (H + 1) + (Gr - f * H) + 1; }   // output row -> its virtual row
  __host__ __device__ static int need_lo(int t, int P) { const int q = TSP * t < P - 1 ? TSP * t : P - 1; return vc_of(q / H) - 1; }
  __host__ __device__ static int need_hi(int t, int P) { const int q = (TSP * (t + 1) < P ? TSP * (t + 1) : P) - 1; return vc_of(q / H) + 1; }
  // rows read between the barriers of steps t and t+1 were requested at the barrier of step t-1 or earlier: need_hi(t+1) < need_lo(t-1) + NR
  static bool ring_ok(int nf) {
    const int P = nf * FP, ns = (P + TSP - 1) / TSP;
    if (need_hi(ns > 1 ? 1 : 0, P) + 1 > NR) return false;        // what is requested before the first step fits the ring
    for (int t = 1; t + 1 < ns; ++t) if (need_hi(t + 1, P) >= need_lo(t - 1, P) + NR) return false;
    return true;
  }
};

static __device__ __forceinline__ void rn_rw_glds4(const float* g_lane, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane, (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// MASKIN (EPI 3 / 4): the relu mask comes as ONE BIT per element (mask_in: a uint16 / uint32 of CO bits per position, written by the forward kernel
// that produced the masked tensor) instead of re-reading the tensor itself — the 16-channel input gradients move 1.2-1.7 GB per launch and sit at
// 3.6-4.0 TB/s; a third / a quarter of that was the mask source.  mask_out (EPI 1 / 5 / 6, may be null): emit that mask for this kernel's output.
template <class G, bool PRE_RELU, int EPI, bool MASKIN>
__global__ __launch_bounds__(512) void rn_rw_kernel(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                                                    const float* aux, float* out, int B, int fpb, const void* __restrict__ mask_in, void* __restrict__ mask_out,
                                                    float* __restrict__ out_relu) {   // out_relu (EPI 1, may be null): a second copy of the output, through a relu
  constexpr int H = G::H, CI = G::CI, CO = G::CO, NT = G::NT, NCO = G::NCO, QPT = G::QPT, NG = G::NG, GF = G::GF, NR = G::NR;
  constexpr bool AUX = EPI == 1 || EPI == 3 || EPI == 4 || EPI == 6, BIAS = EPI == 0 || EPI == 1 || EPI == 5 || EPI == 6;
  constexpr bool MIN = MASKIN && (EPI == 3 || EPI == 4), MOUT = EPI == 1 || EPI == 5 || EPI == 6;
  using MW = typename std::conditional<G::CO == 16, uint16_t, uint32_t>::type;   // CO mask bits of one position
  extern __shared__ __attribute__((aligned(16))) float rw_smem[];
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f0 = blockIdx.x * fpb, nf = min(fpb, B - f0);
  if (nf <= 0) return;
  const int P = nf * G::FP, nsteps = (P + G::TSP - 1) / G::TSP, NV = nf * (H + 1) + 1;
  for (int i = tid; i < G::LDS_FLOATS; i += 512) rw_smem[i] = 0.0f;      // halo cells stay zero for the whole kernel
  __syncthreads();

  int issued = 0;                                                        // virtual rows [0, issued) have been requested
  // A row is copied by ONE wave as NGD unrolled instructions (uniform row pointer + lane, constant granule offsets): the first version dealt
  // (row, granule) pairs round robin with a division, a pointer select and a 64-bit per-lane address for every 256-byte copy — ~30 instructions
  // each, 540 per wave and step against ~700 for the step's arithmetic: 30 % of the kernel (557 -> 396 us on the 16-channel layers without them)
  constexpr int NGD = (G::ROWF + 63) / 64, LASTN = G::ROWF - 64 * (NGD - 1);   // granules that carry data, valid lanes of the last one
  auto issue_row = [&](int v) __attribute__((always_inline)) {
    const int fz = v / (H + 1), yy = v - fz * (H + 1);
    const float* src = (yy == 0 ? rn_rw_zero_row : in + ((size_t)(f0 + fz) * H + (yy - 1)) * G::ROWF) + lane;
    const int pr = v % NR;
    float* dst = rw_smem + (1 + pr * NG) * GF;
#pragma unroll
    for (int g = 0; g < NGD; ++g)
      if (g < NGD - 1 || LASTN == 64 || lane < LASTN) rn_rw_glds4(src + g * 64, dst + g * GF);   // lanes past the row's last pixel: zero halo cells
    if (pr < 2) {                                                         // mirror copy behind the ring
      float* dm = dst + NR * NG * GF;
#pragma unroll
      for (int g = 0; g < NGD; ++g)
        if (g < NGD - 1 || LASTN == 64 || lane < LASTN) rn_rw_glds4(src + g * 64, dm + g * GF);
    }
  };
  auto issue_upto = [&](int lim) __attribute__((always_inline)) {
    lim = min(lim, NV);
    for (int v = issued + wave; v < lim; v += G::NW) issue_row(v);
    issued = max(issued, lim);
  };
  // RW_LDS_RELU (PRE_RELU kernels): the relu in front of the conv is applied ONCE per landed row, in LDS, by the wave that requested the row — between
  // its own vmcnt(0) and the step's barrier, a whole step before the row's first reader — instead of on every fragment (each LDS element is read by
  // nine taps: 144 v_max per 144 MFMAs per wave and step, none of them hidden behind an fp32 MFMA)
  int relu_lo = 0, relu_hi = 0;                                          // rows [relu_lo, relu_hi) were requested by the last issue_upto
  auto relu_rows = [&]() __attribute__((always_inline)) {
    for (int v = relu_lo + wave; v < relu_hi; v += G::NW) {
      const int fz = v / (H + 1), yy = v - fz * (H + 1);
      if (yy == 0) continue;                                             // a zero row
      const int pr = v % NR;
      float* dst = rw_smem + (1 + pr * NG) * GF + lane;
#pragma unroll
      for (int g = 0; g < NGD; ++g)
        if (g < NGD - 1 || LASTN == 64 || lane < LASTN) {
          const float x = fmaxf(dst[g * GF], 0.0f);
          dst[g * GF] = x;
          if (pr < 2) dst[NR * NG * GF + g * GF] = x;                    // its mirror copy
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  issue_upto(G::need_hi(nsteps > 1 ? 1 : 0, P) + 1);
  relu_hi = issued;

  float w[NCO][G::NSTEP];
#pragma unroll
  for (int jc = 0; jc < NCO; ++jc)
#pragma unroll
    for (int s = 0; s < G::NSTEP; ++s) w[jc][s] = W[(size_t)(4 * s + g4) * CO + 16 * jc + r16];
  float bz[NCO];
#pragma unroll
  for (int jc = 0; jc < NCO; ++jc) bz[jc] = BIAS ? bias[16 * jc + r16] : 0.0f;
  const size_t Q0 = (size_t)f0 * G::FP;

  auto sync_and_issue = [&](int t) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // everything requested one step ago has landed (for this wave)
    if constexpr (PRE_RELU && RW_LDS_RELU) relu_rows();
    asm volatile("s_barrier" ::: "memory");                     // ... for every wave; and every wave has left step t-1: rows below need_lo(t) are free
    relu_lo = issued;
    issue_upto(G::need_lo(t, P) + NR);
    relu_hi = issued;
  };

  // FULL: every position of the step exists (all steps but the last): no clamps, no store predicates, element offsets are immediates
  auto step = [&](auto ntl_, auto full_, int t) __attribute__((always_inline)) {
    constexpr int NTL = decltype(ntl_)::value;
    constexpr bool FULL = decltype(full_)::value;
    // A wave owns a span of 16 NT consecutive positions per step; its NT tiles INTERLEAVE over the span (row r of tile j = position NT r + j):
    // a ds_read_b32 is served in two groups of 32 lanes (g4 = 0 / 1 x 16 rows) over 32 banks, and the pixels of one 64-float copy granule share
    // their banks (pixel strides of 16 / 32 floats) — with one pixel per granule and tile, and granules at a pitch of 66 = 2 mod 32, the 32 lanes
    // of a group sit on banks 2 m + g4: conflict-free inside an image row (consecutive positions per tile were 2-way: PMC conflict share 60-67 %)
    static_assert(NTL == NT, "tiles interleave over the wave's span");
    const int qspan = G::TSP * t + wave * (16 * NT);
    int base[NTL][3];
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      uint32_t q = (uint32_t)(qspan + NT * r16 + j);
      if (!FULL) q = min(q, (uint32_t)(P - 1));
#if RW_ABL & 1   // timing build: no address decode (wrong results)
      for (int kw = 0; kw < 3; ++kw) base[j][kw] = (int)((q & 1023u) + kw * 16 + g4);
#else
      const uint32_t Gr = q / H, x = q - Gr * H, f = Gr / H, y = Gr - f * H;   // (unsigned: the divisions by constants are a multiply and a shift)
      const uint32_t s = (f * (H + 1) + y) % NR;                        // slot of the window's first row (virtual row vc - 1)
      const uint32_t p0 = G::GP + s * G::RPX + x - 1;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) { const uint32_t p = p0 + kw; base[j][kw] = (int)((p / G::GP) * GF + (p % G::GP) * CI + g4); }
#endif
    }
    // element e of tile j sits at 32-bit element offset ob[j] + e * CO + 16 jc: one VGPR per tile, the rest immediates (at most 216 M elements per tensor)
    uint32_t ob[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) ob[j] = ((uint32_t)Q0 + (uint32_t)(qspan + NT * 4 * g4 + j)) * CO + r16;   // element e: + e * NT * CO
    float ax[AUX && !MIN ? NTL : 1][NCO][4], ao[EPI == 4 ? NTL : 1][NCO][4];
    uint32_t mw[MIN ? NTL : 1][4];
    if constexpr (AUX) {
      const uint32_t qb = (uint32_t)Q0 + (uint32_t)(qspan + NT * 4 * g4);
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint32_t o = ob[j] + e * NT * CO, qi = qb + j + e * NT;
          if (!FULL) {
            const int q = min(qspan + NT * (4 * g4 + e) + j, P - 1);
            o = ((uint32_t)Q0 + (uint32_t)q) * CO + r16;
            qi = (uint32_t)Q0 + (uint32_t)q;
          }
          if constexpr (MIN) mw[j][e] = (uint32_t)reinterpret_cast<const MW*>(mask_in)[qi];
#pragma unroll
          for (int jc = 0; jc < NCO; ++jc) {
            if constexpr (!MIN) ax[j][jc][e] = aux[o + 16 * jc];
            if constexpr (EPI == 4) ao[j][jc][e] = out[o + 16 * jc];
          }
        }
    }
    rn_f32x4 acc[NTL][NCO];
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int jc = 0; jc < NCO; ++jc) acc[j][jc] = rn_f32x4{0.f, 0.f, 0.f, 0.f};
    // K loop in chunks of 4 channel quads (a tap at CI = 16, half a tap at CI = 32): the fragments of chunk u+1 are read while chunk u is multiplied
    constexpr int QC = 4, CPT = QPT / QC, NCH = 9 * CPT;
    float a[2][NTL][QC];
    auto load_chunk = [&](int u, float (&dst)[NTL][QC]) __attribute__((always_inline)) {
      const int tap = u / CPT, h = u - tap * CPT, kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
      for (int c = 0; c < QC; ++c)
#pragma unroll
        for (int j = 0; j < NTL; ++j) dst[j][c] = rw_smem[base[j][kw] + kh * NG * GF + 4 * (h * QC + c)];
    };
    load_chunk(0, a[0]);
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      if (u == G::SYNC_TAP * CPT) {
        __builtin_amdgcn_sched_barrier(0);
        sync_and_issue(t);
      }
      __builtin_amdgcn_sched_barrier(0);                          // (the scheduler otherwise sinks every read to just before its use)
      if (u + 1 < NCH) load_chunk(u + 1, a[(u + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < QC; ++c)
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          float av = a[u & 1][j][c];
          if (PRE_RELU && !RW_LDS_RELU && !(RW_ABL & 2)) av = fmaxf(av, 0.0f);
#pragma unroll
          for (int jc = 0; jc < NCO; ++jc) acc[j][jc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w[jc][u * QC + c], acc[j][jc], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int jc = 0; jc < NCO; ++jc) {
        uint64_t pos[4];                                              // (MOUT) which lanes hold a positive value in element e
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool valid = FULL || qspan + NT * (4 * g4 + e) + j < P;
          bool on = true;
          if constexpr (EPI == 3 || EPI == 4) {
            if constexpr (MIN) on = ((mw[j][e] >> (16 * jc + r16)) & 1u) != 0u;
            else on = ax[j][jc][e] > 0.0f;
          }
          float v = acc[j][jc][e];
          if (EPI == 0) v = v + bz[jc];
          else if (EPI == 5) v = fmaxf(v + bz[jc], 0.0f);
          else if (EPI == 1) v = (v + bz[jc]) + ax[j][jc][e];
          else if (EPI == 6) v = fmaxf((v + bz[jc]) + ax[j][jc][e], 0.0f);
          else if (EPI == 3) v = on ? v : 0.0f;
          else if (EPI == 4) v = ao[j][jc][e] + (on ? v : 0.0f);
          if (valid) out[ob[j] + e * NT * CO + 16 * jc] = v;
          if constexpr (EPI == 1) { if (out_relu && valid) out_relu[ob[j] + e * NT * CO + 16 * jc] = fmaxf(v, 0.0f); }
          if constexpr (MOUT) pos[e] = __ballot(valid && v > 0.0f);
        }
        if constexpr (MOUT) {
          if (mask_out) {                                             // row 4 g4 + e of the tile: bits [16 g4, +16) of ballot e; lane r16 = e stores it
            const uint64_t mine = r16 == 0 ? pos[0] : (r16 == 1 ? pos[1] : (r16 == 2 ? pos[2] : pos[3]));
            const int q = qspan + NT * (4 * g4 + r16) + j;
            if (r16 < 4 && (FULL || q < P))
              reinterpret_cast<uint16_t*>(mask_out)[(Q0 + (size_t)q) * NCO + jc] = (uint16_t)((mine >> (16 * g4)) & 0xffffu);
          }
        }
      }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (PRE_RELU && RW_LDS_RELU) relu_rows();
  asm volatile("s_barrier" ::: "memory");                       // the rows of the first step and a half are in LDS
  for (int t = 0; t < nsteps; ++t) {
    if (G::TSP * (t + 1) <= P) { step(std::integral_constant<int, NT>{}, std::true_type{}, t); continue; }
    if (G::TSP * t + wave * (16 * NT) < P) step(std::integral_constant<int, NT>{}, std::false_type{}, t);   // short last step: spans that begin inside the block
    else sync_and_issue(t);                                       // nothing to multiply in this (last) step, but the barrier is everybody's
  }
}

template <class G, bool PRE_RELU, int EPI, bool MASKIN = false>
static void rn_rw_launch(const float* in, const float* W, const float* bias, const float* aux, float* out, int B, hipStream_t st,
                         const void* mask_in = nullptr, void* mask_out = nullptr, float* out_relu = nullptr) {
  constexpr int lds = G::LDS_FLOATS * 4;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)rn_rw_kernel<G, PRE_RELU, EPI, MASKIN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  const int fpb = (B + 255) / 256, blocks = (B + fpb - 1) / fpb;
  static int checked_fpb = 0;
  if (checked_fpb != fpb) {
    if (!G::ring_ok(fpb)) { cbm_launch_fail("rn_rw_kernel: row ring of %d slots too small for H=%d TSP=%d at %d frames per block", G::NR, G::H, G::TSP, fpb); return; }
    checked_fpb = fpb;
  }
  hipLaunchKernelGGL((rn_rw_kernel<G, PRE_RELU, EPI, MASKIN>), dim3(blocks), dim3(512), lds, st, in, W, bias, aux, out, B, fpb, mask_in, mask_out, out_relu);
}

// geometry per layer shape: tiles per wave and step, ring slots (checked by ring_ok at launch)
template <int CI, int CO, int H> struct RnRwPick;
template <> struct RnRwPick<16, 16, 42> { using G = RnRwGeom<16, 16, 42, 4, 44>; };
template <> struct RnRwPick<16, 32, 42> { using G = RnRwGeom<16, 32, 42, 4, 44>; };
template <> struct RnRwPick<32, 16, 42> { using G = RnRwGeom<32, 16, 42, 2, 24>; };
template <> struct RnRwPick<32, 32, 21> { using G = RnRwGeom<32, 32, 21, 2, 44>; };
template <> struct RnRwPick<32, 32, 11> { using G = RnRwGeom<32, 32, 11, 2, 82>; };
