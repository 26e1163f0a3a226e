// rnconv.h — direct 3x3 SAME convolutions of the IMPALA-ResNet torso (ppo:149-189) for gfx950.
//
// Why not the implicit-GEMM template: these convs have tiny N (16/32 output channels) and K = 9*Cin, so the im2col
// gather (9 scattered re-reads per input, per-chunk LDS stores and barriers) costs more than the math.  Here a block
// owns a strip of R image rows (or NF whole small frames), copies the (R+2)-row input slab ONCE into LDS as channel
// planes with a zero pad column per row, and then every tap of the 3x3 window is the SAME slab read at a constant
// offset:  with row pitch WP = W+1 (the pad column of row r is also the left pad of row r+1) the input of flat
// output position q for tap (kh,kw) sits at slab index q + kh*WP + kw.  The K loop is therefore barrier-free, all
// LDS addresses are one base register + immediates, and the weights stay in LDS for the whole block.
//
// Numerics (forward): per output, k = (kh,kw,ci) ascending fp32 fmaf chain from 0, bias added afterwards — identical
// to the oracle and to the implicit-GEMM kernels this replaces; pad taps contribute fma(0,w,acc).
//   CO == 32: v_mfma_f32_32x32x2_f32  (A: lane -> position l%32, k = 2j + l/32)
//   CO == 16: v_mfma_f32_16x16x4_f32  (A: lane -> position l%16, k = 4j + l/16)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

typedef float rn_f32x4 __attribute__((ext_vector_type(4)));
typedef float rn_f32x16 __attribute__((ext_vector_type(16)));

constexpr int rn_round_up_mod(int v, int m, int r) {  // smallest x >= v with x % m == r
  int x = v - (v % m) + r;
  return x < v ? x + m : x;
}

template <int CI_, int CO_, int H_, int R_, int NF_, int NW_ = 4>
struct RnGeom {
  static constexpr int CI = CI_, CO = CO_, H = H_, R = R_, NF = NF_, NW = NW_, NTHR = 64 * NW_;   // NW waves share one slab + weight copy
  static constexpr int WP = H + 1;
  static constexpr bool M16 = CO == 16;
  static constexpr int TP = M16 ? 16 : 32;                        // positions per MFMA tile
  static constexpr int KK = M16 ? 4 : 2;                          // k per MFMA
  static constexpr int OROWS = NF == 1 ? R : NF * (H + 1);        // output rows per block (incl. inter-frame gap rows)
  static constexpr int SROWS = NF == 1 ? R + 2 : NF * (H + 1) + 1;  // slab rows
  static constexpr int NQ = OROWS * WP;
  static constexpr int NT = (NQ + TP - 1) / TP;
  static constexpr int NTW = (NT + NW - 1) / NW;                  // tiles per wave
  static constexpr int S = NT * TP + 2 * WP + 3;                  // slab positions that may be read
  // channel-plane placement: M16 reads planes (4j..4j+3) with 16 lanes each -> planes of a pair must sit 16 banks apart
  static constexpr int PLH = M16 ? rn_round_up_mod(S, 32, 16) : 0;
  static constexpr int PL = M16 ? rn_round_up_mod(PLH + S, 16, 4) : rn_round_up_mod(S, 8, 2);  // M16: pitch of a plane PAIR
  static constexpr int NPL = CI < 4 ? 4 : CI;
  static constexpr int SLAB = M16 ? (NPL / 2) * PL : NPL * PL;    // floats
  static constexpr int WSZ = 9 * CI * CO;
  static constexpr int STRIPS = NF == 1 ? (H + R - 1) / R : 1;    // strips per frame
  static constexpr int LDS_BYTES = (SLAB + WSZ) * 4;
  static constexpr int OCC = 160 * 1024 / LDS_BYTES >= 4 ? 4 : (160 * 1024 / LDS_BYTES < 1 ? 1 : 160 * 1024 / LDS_BYTES);  // blocks per CU by LDS
  static constexpr int MINW = OCC * NW / 4 >= 4 ? 4 : (OCC * NW / 4 >= 2 ? 2 : 1);   // waves per SIMD the LDS footprint allows -> VGPR budget
  __host__ __device__ static constexpr int poff(int p) { return M16 ? (p >> 1) * PL + (p & 1) * PLH : p * PL; }
  static int blocks(int B) { return NF == 1 ? B * STRIPS : (B + NF - 1) / NF; }
};

// EPI: 0 out = v + bias            (forward)
//      1 out = v + bias + aux      (forward, residual add)
//      2 out = v                   (dgrad)
//      3 out = aux > 0 ? v : 0     (dgrad through the relu in front of the conv)
//      4 out += aux > 0 ? v : 0    (dgrad joined with the residual path, in place)
//      5 out = relu(v + bias)      (forward, for an output that is only ever read through a relu: the first conv of a residual block)
//      6 out = relu(v + bias + aux) (forward, residual add of the torso's LAST block: only the dense layer — through a relu — and relu masks read it)
// Blocks are PERSISTENT over strips (grid = what fits on the chip): the global loads of strip s+1 — its input slab and, for EPI 1/3/4, the
// residual / mask values its epilogue needs — are issued before the MFMA sweep of strip s and land during it; the first build ran one strip
// per block as stage -> barrier -> multiply -> store, every phase exposed (MFMA phase ~40 % of a block's life, 0.41-0.46 of the peak).
template <class G, bool PRE_RELU, int EPI>
__global__ __launch_bounds__(G::NTHR, G::MINW) void rn_conv_kernel(const float* in, const float* W, const float* bias, const float* aux, float* out, int B,
                                                                  int nstrips) {
  constexpr int H = G::H, CI = G::CI, CO = G::CO, WP = G::WP, TP = G::TP, NTW = G::NTW, NT = G::NT;
  constexpr int NE = G::M16 ? 4 : 16;                                   // accumulator elements per lane per tile
  constexpr bool AUX = EPI == 1 || EPI == 3 || EPI == 4 || EPI == 6;
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  float* slab = rn_smem;
  float* Wl = rn_smem + G::SLAB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int v = tid; v < G::WSZ / 4; v += G::NTHR) reinterpret_cast<float4*>(Wl)[v] = reinterpret_cast<const float4*>(W)[v];
  // pad column of every slab row (index sr*WP + WP) and the leading pad (index 0): written once, the strip copies never touch them
  for (int v = tid; v < (G::SROWS + 1) * G::NPL; v += G::NTHR) {
    const int p = v % G::NPL, sr = v / G::NPL;
    slab[G::poff(p) + sr * WP] = 0.0f;
  }
  // ---- per-thread index tables, decoded ONCE: which slab elements this thread copies and which outputs it stores is the same for every strip up
  // to the strip's origin (the decode is a dozen integer divisions per element; done per strip it was ~400 VALU instructions per thread against
  // ~100 MFMAs per wave — and VALU work takes the fp32 matrix pipe's slots on this chip)
  constexpr int NVX = G::SROWS * H * (CI / 4), NIX = (NVX + G::NTHR - 1) / G::NTHR;
  float4 rx[NIX];
  int cp_col[NIX], cp_dst[NIX], cp_row[NIX];      // source column part (floats), slab index of channel 4g, slab row (NF = 1) / frame<<8 | row+1 (NF = 2)
#pragma unroll
  for (int it = 0; it < NIX; ++it) {
    const int v = min(tid + G::NTHR * it, NVX - 1);
    const int g = v % (CI / 4), pix = v / (CI / 4), c = pix % H, sr = pix / H;
    cp_col[it] = c * CI + 4 * g;
    cp_dst[it] = G::poff(4 * g) + 1 + sr * WP + c;
    cp_row[it] = G::NF == 1 ? sr : (((sr / (H + 1)) << 8) | (sr % (H + 1)));
  }
  const int li = G::M16 ? (lane & 15) : (lane & 31), kq = G::M16 ? (lane >> 4) : (lane >> 5);
  // outputs: element e of tile i -> offset relative to the strip origin (bits 0..19), its strip row (NF = 1) or frame<<6 | row (NF = 2) in bits
  // 20..30, bit 31 = never stored (pad column, row beyond the strip, tile beyond the strip)
  uint32_t oc[NTW][NE];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int tile = wave + G::NW * i;
      const int q = tile * TP + (G::M16 ? 4 * kq + e : (e & 3) + 8 * (e >> 2) + 4 * kq);
      const int orow = q / WP, c = q - orow * WP;
      const bool dead = tile >= NT || c >= H || (G::NF == 1 ? orow >= G::R : (orow % (H + 1)) >= H);
      const int fs = G::NF == 1 ? 0 : orow / (H + 1), y = G::NF == 1 ? orow : orow % (H + 1);
      const uint32_t rel = (uint32_t)(((fs * H + y) * H + c) * CO + li);
      oc[i][e] = dead ? 0x80000000u : (rel | ((uint32_t)(G::NF == 1 ? orow : ((fs << 6) | y)) << 20));
    }
  static_assert(((G::NF * H + H) * H + H) * CO < (1 << 20), "relative output offsets fit 20 bits");
  auto where = [&](int st, int& b0, int& y0) {
    if (G::NF == 1) { b0 = st / G::STRIPS; y0 = (st - b0 * G::STRIPS) * G::R; }
    else { b0 = st * G::NF; y0 = 0; }
  };
  auto fetch = [&](int b0, int y0) {
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
      int f, y;
      if (G::NF == 1) { f = b0; y = y0 + cp_row[it] - 1; }
      else { f = b0 + (cp_row[it] >> 8); y = (cp_row[it] & 255) - 1; }
      rx[it] = *reinterpret_cast<const float4*>(in + (size_t)((min(f, B - 1) * H + min(max(y, 0), H - 1)) * H * CI + cp_col[it]));
    }
  };
  auto commit = [&](int b0, int y0) {
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
      if (NVX % G::NTHR != 0 && tid + G::NTHR * it >= NVX) break;
      int f, y;
      if (G::NF == 1) { f = b0; y = y0 + cp_row[it] - 1; }
      else { f = b0 + (cp_row[it] >> 8); y = (cp_row[it] & 255) - 1; }
      const bool ok = y >= 0 && y < H && f < B;
      const float e[4] = {rx[it].x, rx[it].y, rx[it].z, rx[it].w};
      float* d = slab + cp_dst[it];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float val = ok ? e[q] : 0.0f;
        if (PRE_RELU) val = fmaxf(val, 0.0f);
        d[G::poff(q)] = val;                       // poff(4g + q) - poff(4g) = poff(q): plane pairs / planes are a constant pitch apart
      }
    }
  };
  // offset of output (i, e) for the strip at (b0, y0), or -1
  auto out_index = [&](uint32_t code, int b0, int y0) -> int {
    if ((int)code < 0) return -1;
    const int rw = (int)(code >> 20);
    if (G::NF == 1) { if (y0 + rw >= H) return -1; }
    else { if (b0 + (rw >> 6) >= B) return -1; }
    return (b0 * H + y0) * H * CO + (int)(code & 0xFFFFFu);
  };
  using AccT = typename std::conditional<G::M16, rn_f32x4, rn_f32x16>::type;

  int s = blockIdx.x, b0 = 0, y0 = 0;
  if (s < nstrips) { where(s, b0, y0); fetch(b0, y0); }
  const float bz = (EPI == 0 || EPI == 1 || EPI == 5 || EPI == 6) ? bias[li] : 0.0f;
  for (; s < nstrips; s += gridDim.x) {
    __syncthreads();            // the previous strip's sweep is done with the slab (and the weights / pads are staged)
    commit(b0, y0);
    const int cb0 = b0, cy0 = y0;
    __syncthreads();
    const int sn = s + gridDim.x;
    if (sn < nstrips) { where(sn, b0, y0); fetch(b0, y0); }
    float ax[AUX ? NTW : 1][AUX ? NE : 1], ao[EPI == 4 ? NTW : 1][EPI == 4 ? NE : 1];
    if constexpr (AUX) {        // residual / mask values (EPI 4: and the running gradient) of THIS strip's outputs: requested before the sweep, used after it
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int o = out_index(oc[i][e], cb0, cy0);
          ax[i][e] = aux[o < 0 ? 0 : o];
          if constexpr (EPI == 4) ao[i][e] = out[o < 0 ? 0 : o];
        }
    }
    AccT acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[i][e] = 0.0f;
    // tiles this wave really has in this strip (wave-uniform): tile t exists iff t * TP < rows * WP.  The first build multiplied all NTW slots
    // of every wave — 24 tile slots for the 19 tiles of a 7-row strip of the 42x42 layers: a fifth of the matrix pipe's work was on positions
    // nobody stores (a co-resident block cannot use slots that are busy with garbage)
    int rows = G::NF == 1 ? min(G::R, H - cy0) : G::OROWS;
    const int ntl = max(0, min(NTW, ((rows * WP + TP - 1) / TP - wave + G::NW - 1) / G::NW));
    auto sweep = [&](auto NTL_) __attribute__((always_inline)) {
      constexpr int NTL = decltype(NTL_)::value;
      if constexpr (G::M16) {
        const float* abase = slab + (kq >> 1) * G::PL + (kq & 1) * G::PLH + wave * TP + li;   // plane of k = 4j + kq
        const float* bbase = Wl + kq * CO + li;
        // one tap per (rolled) iteration: bounds the scheduling region and with it the number of hoisted LDS reads (VGPRs)
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
          const int kh = (t * 11) >> 5, off = kh * WP + (t - 3 * kh);
          const float* bt = bbase + t * CI * CO;
#pragma unroll
          for (int j = 0; j < (CI < 4 ? 1 : CI / 4); ++j) {
            const float bv = bt[4 * j * CO];
            const float* ap = abase + 2 * j * G::PL + off;
#pragma unroll
            for (int i = 0; i < NTL; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[i * G::NW * TP], bv, acc[i], 0, 0, 0);
          }
        }
      } else {
        const float* abase = slab + kq * G::PL + wave * TP + li;
        const float* bbase = Wl + kq * CO + li;
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
          const int kh = (t * 11) >> 5, off = kh * WP + (t - 3 * kh);
          const float* bt = bbase + t * CI * CO;
#pragma unroll
          for (int j = 0; j < CI / 2; ++j) {
            const float bv = bt[2 * j * CO];
            const float* ap = abase + 2 * j * G::PL + off;
#pragma unroll
            for (int i = 0; i < NTL; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[i * G::NW * TP], bv, acc[i], 0, 0, 0);
          }
        }
      }
    };
    if (ntl == NTW) sweep(std::integral_constant<int, NTW>{});
    else if (NTW > 1 && ntl == NTW - 1) sweep(std::integral_constant<int, (NTW > 1 ? NTW - 1 : 1)>{});
    else if (NTW > 2 && ntl == NTW - 2) sweep(std::integral_constant<int, (NTW > 2 ? NTW - 2 : 1)>{});
    else if (ntl > 0) sweep(std::integral_constant<int, NTW>{});   // (deeper shortfalls do not occur with the shipped geometries; correct either way)
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int o = out_index(oc[i][e], cb0, cy0);
        if (o < 0) continue;
        float v = acc[i][e];
        if (EPI == 0) v = v + bz;
        else if (EPI == 5) v = fmaxf(v + bz, 0.0f);
        else if (EPI == 1) v = (v + bz) + ax[i][e];
        else if (EPI == 6) v = fmaxf((v + bz) + ax[i][e], 0.0f);
        else if (EPI == 3) v = ax[i][e] > 0.0f ? v : 0.0f;
        else if (EPI == 4) v = ao[i][e] + (ax[i][e] > 0.0f ? v : 0.0f);
        out[o] = v;
      }
  }
}

template <class G, bool PRE_RELU, int EPI>
static void rn_conv_launch(const float* in, const float* W, const float* bias, const float* aux, float* out, int B, hipStream_t st) {
  constexpr size_t lds = (size_t)(G::SLAB + G::WSZ) * sizeof(float);
  static_assert(lds <= 160 * 1024, "slab + weights exceed the 160 KB LDS of a gfx950 CU");
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)rn_conv_kernel<G, PRE_RELU, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int nstrips = G::blocks(B);
  int per_cu = G::OCC;                                              // blocks per CU by LDS, capped by the 32 wave slots
  if (per_cu * G::NW > 32) per_cu = 32 / G::NW;
  const int nb = nstrips < 256 * per_cu ? nstrips : 256 * per_cu;
  hipLaunchKernelGGL((rn_conv_kernel<G, PRE_RELU, EPI>), dim3(nb), dim3(G::NTHR), lds, st, in, W, bias, aux, out, B, nstrips);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) fprintf(stderr, "rn_conv launch failed: %s (threads %d, lds %zu)\n", hipGetErrorString(e), G::NTHR, lds);
}

// ---- first conv of the torso fused with its max_pool(3,3) stride 2 SAME (ppo:158-166): the 84x84x16 conv output (1.7 GB per
// 3840-frame minibatch) never reaches HBM.  A block produces PR pooled rows: it convolves the 2*PR+1 rows they cover into LDS
// (+bias), then pools them (same (kh,kw) scan, first max wins -> same arg-max bytes as rn_pool_fwd_kernel).  One conv row per
// strip is computed twice (7 rows for 6 unique); nothing downstream reads the un-pooled tensor (pool backward uses the arg-max).
template <int PR>
struct RnPool0Geom {
  static constexpr int H = 84, HP = 42, CO = 16, R = 2 * PR + 1;
  using G = RnGeom<4, 16, 84, R, 1, 4>;
  static constexpr int STRIPS = (HP + PR - 1) / PR;
  static constexpr int CV = R * H * CO;                           // conv rows held in LDS
  static constexpr int LDS_FLOATS = G::SLAB + G::WSZ + CV;
};
// Blocks are persistent over strips: the bytes of the next strip (3 uint32 per thread) are requested before the sweep of the current one.
template <int PR>
__global__ __launch_bounds__(256, 2) void rn_conv0_pool_kernel(const uint8_t* obs, const int32_t* idx, const float* W, const float* bias, float* pooled,
                                                               uint8_t* pidx, int B, int nstrips, uint16_t* mask) {   // mask: null, or 16 bits per pooled position (value > 0)
  using PG = RnPool0Geom<PR>;
  using G = typename PG::G;
  constexpr int H = 84, HP = 42, CO = 16, WP = G::WP, TP = G::TP, NTW = G::NTW, NT = G::NT, R = PG::R;
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  float* slab = rn_smem;
  float* Wl = rn_smem + G::SLAB;
  float* cv = Wl + G::WSZ;                                        // [R][H][CO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int v = tid; v < G::WSZ / 4; v += 256) reinterpret_cast<float4*>(Wl)[v] = reinterpret_cast<const float4*>(W)[v];
  for (int v = tid; v < (G::SROWS + 1) * 4; v += 256) {           // pad columns: written once
    const int p = v % 4, sr = v / 4;
    slab[G::poff(p) + sr * WP] = 0.0f;
  }
  // this thread's share of a strip: uint32 v = (plane p, slab row sr, 4 columns cq) — decoded once
  constexpr int NV = G::SROWS * (H / 4) * 4, NI = (NV + 255) / 256;
  uint32_t ru[NI];
  int usrc[NI], udst[NI], urow[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int v = min(tid + 256 * it, NV - 1);
    const int cq = v % (H / 4), t = v / (H / 4), sr = t % G::SROWS, p = t / G::SROWS;
    usrc[it] = p * H * H + 4 * cq; udst[it] = G::poff(p) + 1 + sr * WP + 4 * cq; urow[it] = sr;
  }
  auto fetch = [&](int st) {
    const int b0 = st / PG::STRIPS, y0 = 2 * ((st % PG::STRIPS) * PR);
    const uint8_t* fr = obs + (size_t)(idx ? idx[b0] : b0) * CBM_FRAME;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int y = y0 + urow[it] - 1;
      ru[it] = *reinterpret_cast<const uint32_t*>(fr + min(max(y, 0), H - 1) * H + usrc[it]);
    }
  };
  const int li = lane & 15, kq = lane >> 4;
  const float bz = bias[li];
  // where element e of tile i goes in cv (conv row-major [R][H][CO]), decoded once: 0xffff = pad column / row beyond the strip / tile beyond the strip
  // (per strip it was a division and two compares per stored value: ~400 VALU instructions per wave against 90 MFMAs)
  uint32_t cvo[NTW][2];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int tile = wave + 4 * i, q = tile * TP + 4 * kq + e, orow = q / WP, c = q - orow * WP;
      const uint32_t o = (tile < NT && c < H && orow < R) ? (uint32_t)((orow * H + c) * CO + li) : 0xffffu;
      static_assert(R * H * CO < 0xffff, "cv offsets fit 16 bits");
      if (e & 1) cvo[i][e >> 1] |= o << 16; else cvo[i][e >> 1] = o;
    }
  int s = blockIdx.x;
  if (s < nstrips) fetch(s);
  for (; s < nstrips; s += gridDim.x) {
    const int b0 = s / PG::STRIPS, p0 = (s % PG::STRIPS) * PR, y0 = 2 * p0;   // pad_lo = 0 for 84 -> 42
    __syncthreads();            // the previous strip's pooling pass is done with cv, its sweep with the slab
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      if (NV % 256 != 0 && tid + 256 * it >= NV) break;
      const int y = y0 + urow[it] - 1;
      const bool ok = y >= 0 && y < H;
      float* d = slab + udst[it];
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = ok ? cbm_u8_unit((uint8_t)((ru[it] >> (8 * q)) & 0xffu)) : 0.0f;
    }
    __syncthreads();
    if (s + (int)gridDim.x < nstrips) fetch(s + gridDim.x);
    {
      rn_f32x4 acc[NTW];
#pragma unroll
      for (int i = 0; i < NTW; ++i) acc[i] = rn_f32x4{0.f, 0.f, 0.f, 0.f};
      const float* abase = slab + (kq >> 1) * G::PL + (kq & 1) * G::PLH + wave * TP + li;
      const float* bbase = Wl + kq * CO + li;
#pragma unroll 1
      for (int t = 0; t < 9; ++t) {
        const int kh = (t * 11) >> 5, off = kh * WP + (t - 3 * kh);
        const float bv = bbase[t * 4 * CO];
        const float* ap = abase + off;
#pragma unroll
        for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[i * 4 * TP], bv, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t o = (cvo[i][e >> 1] >> (16 * (e & 1))) & 0xffffu;
          if (o != 0xffffu) cv[o] = acc[i][e] + bz;
        }
    }
    __syncthreads();
    // max_pool over the rows held in LDS
    for (int i = tid; i < PR * HP * (CO / 4); i += 256) {
      const int c4 = i % (CO / 4), ow = (i / (CO / 4)) % HP, ohl = i / ((CO / 4) * HP), oh = p0 + ohl;
      if (oh >= HP) continue;
      float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      uint32_t bq[4] = {0u, 0u, 0u, 0u};                            // arg-max per channel (packed into bytes at the end: 3 instead of 5 VALU per compare)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ih = oh * 2 + kh, iw = ow * 2 + kw;
          if (ih >= H || iw >= H) continue;
          const float4 v = *reinterpret_cast<const float4*>(cv + ((ih - y0) * H + iw) * CO + 4 * c4);
          const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool gt = e[q] > best[q];
            best[q] = gt ? e[q] : best[q];
            bq[q] = gt ? (uint32_t)(kh * 3 + kw) : bq[q];
          }
        }
      const uint32_t bi = bq[0] | (bq[1] << 8) | (bq[2] << 16) | (bq[3] << 24);
      const size_t o = (((size_t)(b0 * HP + oh) * HP + ow) * (CO / 4) + c4) * 4;
      *reinterpret_cast<float4*>(pooled + o) = make_float4(best[0], best[1], best[2], best[3]);
      *reinterpret_cast<uint32_t*>(pidx + o) = bi;
      if (mask) {   // the four lanes of a position are consecutive and take the `continue` above together
        uint32_t wv = ((best[0] > 0.0f ? 1u : 0u) | (best[1] > 0.0f ? 2u : 0u) | (best[2] > 0.0f ? 4u : 0u) | (best[3] > 0.0f ? 8u : 0u)) << (4 * c4);
        wv |= __shfl_xor(wv, 1);
        wv |= __shfl_xor(wv, 2);
        if (c4 == 0) mask[o / 16] = (uint16_t)wv;
      }
    }
  }
}
template <int PR>
static void rn_conv0_pool_launch(const uint8_t* obs, const int32_t* idx, const float* W, const float* bias, float* pooled, uint8_t* pidx, int B,
                                 hipStream_t st, uint16_t* mask = nullptr) {
  using PG = RnPool0Geom<PR>;
  constexpr size_t lds = (size_t)PG::LDS_FLOATS * sizeof(float);
  static_assert(lds <= 160 * 1024, "conv0+pool strip exceeds LDS");
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)rn_conv0_pool_kernel<PR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  const int nstrips = B * PG::STRIPS;
  const int per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : (int)(160 * 1024 / lds);
  const int nb = nstrips < 256 * per_cu ? nstrips : 256 * per_cu;
  hipLaunchKernelGGL((rn_conv0_pool_kernel<PR>), dim3(nb), dim3(256), lds, st, obs, idx, W, bias, pooled, pidx, B, nstrips, mask);
}

// ---- round 6: the same layer with the max_pool done ON THE ACCUMULATORS (learner batches).  rn_conv0_pool_kernel above writes every conv value to LDS
// (four 4-way-conflicting ds_write_b32 per tile), re-reads each one 2.25 times as float4 in a separate pooling pass behind a block barrier, and computes
// one conv row in seven twice: 585 us per 3840-frame minibatch for 198 us of MFMA time (0.34).  Here ONE WAVE owns ONE FRAME and nothing is shared:
//   * a tile is 4 rows x 4 columns of conv outputs (A operand: lane -> position (li >> 2, li & 3)), so an accumulator lane holds four consecutive
//     COLUMNS of one row for its channel: the horizontal 3-max of both pooled columns of the tile (columns 4ct..4ct+2 and 4ct+2..4ct+4) is in-lane
//     arithmetic — the one value from the next column tile is the same lane's accumulator of that tile, computed one tile ahead;
//   * the vertical 3-max goes through 1 KB of wave-private LDS (one ds_write_b128, three ds_read_b64 per tile).  Rows 4g, 4g+1, 4g+2 give pooled row
//     2g (lanes 0-31); rows 4g+2, 4g+3 give the first two thirds of pooled row 2g+1, which stay in REGISTERS (lanes 32-63, one pair per column tile:
//     the 21-tile sweep is unrolled) until the next row group supplies row 4g+4 — 84 = 21 x 4 rows and columns: no conv value is computed twice, no
//     tile slot is empty;
//   * the scan order of the reference's max_pool (first maximum in (kh, kw) order wins, ppo:158-166 / rn_pool_fwd_kernel) is kept by merging in that
//     order with strict compares: same values, same arg-max bytes, same mask bits as the kernel above;
//   * the input rows of a group (4 new + 2 old, 6 x 84 x 4 bytes) are requested before the previous group's sweep and converted after it; blocks of
//     three waves (9.8 KB of LDS per wave), five per CU: the 3840 frames of a learner minibatch are 15 waves on each of the 256 CUs, all resident at once;
//     no block barrier anywhere.
// What bounds it now (timing builds, profiles/r06_resnet_sinks.txt): the 441 x 9 MFMAs of a frame alone are 227 us per minibatch, everything else alone
// 231 us, together 378 — an fp32 MFMA does not overlap VALU work on this chip, and a tile carries ~33 VALU instructions.
#ifndef RN_P0_ABL   // timing builds only (tools/variants.sh): 1 no global stores, 2 no LDS exchange, 4 no MFMAs, 8 no staging
#define RN_P0_ABL 0
#endif
#ifndef RN_P0_DEPTH
#define RN_P0_DEPTH 2
#endif
struct RnPool0Reg {
  static constexpr int H = 84, HP = 42, CO = 16, NCT = 21, NG = 21, NW = 3;
  static constexpr int RP = 88;                 // slab row pitch: 4 zero cells, then the 84 columns (the right pad is the next row's first cell)
  static constexpr int PL = 548;                // plane pitch >= 6 * 88 + 4, = 4 mod 32: the 4 planes x 4 rows x 4 columns of a fragment read sit on 32 banks per half wave
  static constexpr int SLAB = 4 * PL, HB = 64 * 4, WAVE_FLOATS = SLAB + HB;
  static_assert(PL % 32 == 4 && PL >= 6 * RP + 4 && PL % 4 == 0, "plane pitch");
};
__global__ __launch_bounds__(64 * RnPool0Reg::NW, 4) void rn_conv0_pool_reg_kernel(const uint8_t* __restrict__ obs, const int32_t* __restrict__ idx,
                                                                                  const float* __restrict__ W, const float* __restrict__ bias,
                                                                                  float* __restrict__ pooled, uint8_t* __restrict__ pidx, int B,
                                                                                  uint16_t* __restrict__ mask, float* __restrict__ pooled_relu) {
  using G = RnPool0Reg;
  constexpr int H = G::H, HP = G::HP, RP = G::RP, PL = G::PL;
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x * G::NW + wave;
  if (b >= B) return;                                             // (no block barrier below)
  float* slab = rn_smem + wave * G::WAVE_FLOATS;
  float4* hb4 = reinterpret_cast<float4*>(slab + G::SLAB);        // [row kq][channel li] -> (max j=0, kw j=0, max j=1, kw j=1)
  const float2* hb2 = reinterpret_cast<const float2*>(slab + G::SLAB);
  for (int v = lane; v < G::SLAB / 4; v += 64) reinterpret_cast<float4*>(slab)[v] = make_float4(0.f, 0.f, 0.f, 0.f);   // pad cells stay zero
  const uint8_t* fr = obs + (size_t)(idx ? idx[b] : b) * CBM_FRAME;
  float bv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) bv[t] = W[(t * 4 + kq) * 16 + li];
  const float bz = bias[li];
  // slab rows 0..5 of group g = input rows 4g-1 .. 4g+4.  Copy it (0..7) moves rows 3 (it & 1) .. + 2 of plane it >> 1: lane -> (row lane / 21, four columns
  // lane % 21), decoded once; lane 63 idles.  Rows outside the frame are zero BYTES (0 / 255 = 0.0f exactly)
  constexpr int NI = 8;
  uint32_t ru[NI];
  const int sr = min(lane / 21, 2), scq = lane - 21 * (lane / 21);
  const uint8_t* fsrc = fr + sr * H + 4 * scq;
  float* sdst = slab + sr * RP + 4 + 4 * scq;
  auto fetch = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int y = 4 * g - 1 + 3 * (it & 1) + sr;
      const uint32_t w = *reinterpret_cast<const uint32_t*>(fsrc + (it >> 1) * (H * H) + (min(max(y, 0), H - 1) - sr) * H);
      ru[it] = (y >= 0 && y < H) ? w : 0u;
    }
  };
  auto commit = [&](int g) __attribute__((always_inline)) {
    if (lane < 63) {
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        float4 o;
        o.x = cbm_u8_unit(ru[it] & 0xffu);
        o.y = cbm_u8_unit((ru[it] >> 8) & 0xffu);
        o.z = cbm_u8_unit((ru[it] >> 16) & 0xffu);
        o.w = cbm_u8_unit(ru[it] >> 24);
        *reinterpret_cast<float4*>(sdst + (it >> 1) * PL + 3 * (it & 1) * RP) = o;
      }
    }
  };
  // A fragment of tap (kh, kw) of column tile ct: plane kq, slab row (li >> 2) + kh, column 4 ct + (li & 3) + kw - 1 -> one base + immediates
  const float* abase = slab + kq * PL + (li >> 2) * RP + 3 + (li & 3);
  const bool hi = kq >= 2;                                        // lanes 32-63 finish pooled row 2g-1, lanes 0-31 produce pooled row 2g
  const int j = kq & 1;                                           // pooled column 2 ct + j
  const int ra = hi ? 2 : 0, rc = hi ? 0 : 2;                     // (rb = ra + 1)
  const float2* hA = hb2 + ((ra * 16 + li) * 2 + j);
  const float2* hC = hb2 + ((rc * 16 + li) * 2 + j);
  float cv_[G::NCT];                                              // rows 4g+2, 4g+3 of pooled row 2g+1: running maximum and its arg (lanes 32-63)
  uint32_t ck_[(G::NCT + 7) / 8] = {0u, 0u, 0u};                 // (4 bits per column tile)
#pragma unroll
  for (int ct = 0; ct < G::NCT; ++ct) cv_[ct] = 0.0f;
  const size_t fb = (size_t)b * (HP * HP * 16);
  const int lc = (j - (hi ? HP : 0)) * 16 + li;                   // element offset of this lane's output relative to (row 2g, column 2ct)
  auto tile_mfma = [&](int ct, rn_f32x4& acc) __attribute__((always_inline)) {
    acc = rn_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int kh = t / 3, kw = t - 3 * kh;
      if (RN_P0_ABL & 4) acc[t & 3] = __uint_as_float(__float_as_uint(acc[t & 3]) ^ __float_as_uint(abase[kh * RP + kw + 4 * ct]));
      else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(abase[kh * RP + kw + 4 * ct], bv[t], acc, 0, 0, 0);
    }
  };
  fetch(0);
#pragma unroll 1
  for (int g = 0; g < G::NG; ++g) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!(RN_P0_ABL & 8) || g == 0) commit(g);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (g + 1 < G::NG && !(RN_P0_ABL & 8)) fetch(g + 1);
    const long ob = (long)fb + g * (2 * HP * 16) + lc;            // + ct * 32
    float* po = pooled + ob;
    float* pr = pooled_relu + ob;                                 // (may be null: relu(pooled), what the first residual block's first conv reads)
    uint8_t* pi = pidx + ob;
    uint16_t* pm = mask + (ob >> 4);                              // (li == 0 lanes only)
    const bool live = !(hi && g == 0);                            // there is no pooled row -1
    // RN_P0_DEPTH tiles are multiplied ahead of the one being pooled: their 9-step chains are independent of each other
    rn_f32x4 acc[RN_P0_DEPTH + 1];
#pragma unroll
    for (int d = 0; d < RN_P0_DEPTH; ++d) tile_mfma(d, acc[d]);
    float a0 = acc[0][0] + bz;
#pragma unroll
    for (int ct = 0; ct < G::NCT; ++ct) {
      if (ct + RN_P0_DEPTH < G::NCT) tile_mfma(ct + RN_P0_DEPTH, acc[RN_P0_DEPTH]);
      const float a1 = acc[0][1] + bz, a2 = acc[0][2] + bz, a3 = acc[0][3] + bz;
      const float x4 = ct + 1 < G::NCT ? acc[1][0] + bz : -INFINITY;
      float m0 = a0, m1 = a2;
      uint32_t k0 = 0u, k1 = 0u;
      if (a1 > m0) { m0 = a1; k0 = 1u; }
      if (a2 > m0) { m0 = a2; k0 = 2u; }
      if (a3 > m1) { m1 = a3; k1 = 1u; }
      if (x4 > m1) { m1 = x4; k1 = 2u; }
      if (!(RN_P0_ABL & 2)) hb4[lane] = make_float4(m0, __uint_as_float(k0), m1, __uint_as_float(k1));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float2 qa, qb, qc;                                          // rows ra, ra + 1 (16 channels x 2 pairs further), rc
      if (RN_P0_ABL & 2) { qa = make_float2(m0, __uint_as_float(k0)); qb = make_float2(m1, __uint_as_float(k1)); qc = make_float2(a3, __uint_as_float(k0 ^ k1)); }
      else { qa = hA[0]; qb = hA[32]; qc = hC[0]; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float mv = qa.x;
      uint32_t mk = __float_as_uint(qa.y);
      if (qb.x > mv) { mv = qb.x; mk = __float_as_uint(qb.y) + 3u; }
      float fv = hi ? cv_[ct] : mv;
      uint32_t fk = hi ? (ck_[ct >> 3] >> (4 * (ct & 7))) & 15u : mk;
      cv_[ct] = mv;
      ck_[ct >> 3] = (ck_[ct >> 3] & ~(15u << (4 * (ct & 7)))) | (mk << (4 * (ct & 7)));
      if (qc.x > fv) { fv = qc.x; fk = __float_as_uint(qc.y) + 6u; }
      const unsigned long long bal = __ballot(fv > 0.0f);
      if ((RN_P0_ABL & 1) ? (live && fv == 123.456f && fk == 77u) : live) {
        po[ct * 32] = fv;
        if (pooled_relu) pr[ct * 32] = fmaxf(fv, 0.0f);
        pi[ct * 32] = (uint8_t)fk;
        if (mask && li == 0) pm[ct * 2] = (uint16_t)(bal >> (16 * kq));
      }
      a0 = x4;
#pragma unroll
      for (int d = 0; d < RN_P0_DEPTH; ++d) acc[d] = acc[d + 1];
    }
  }
  // pooled row 41: rows 82 and 83 only (row 84 is padding)
  if (hi) {
    const long ob = (long)fb + G::NG * (2 * HP * 16) + lc;
#pragma unroll
    for (int ct = 0; ct < G::NCT; ++ct) {
      const unsigned long long bal = __ballot(cv_[ct] > 0.0f);
      pooled[ob + ct * 32] = cv_[ct];
      if (pooled_relu) pooled_relu[ob + ct * 32] = fmaxf(cv_[ct], 0.0f);
      pidx[ob + ct * 32] = (uint8_t)((ck_[ct >> 3] >> (4 * (ct & 7))) & 15u);
      if (mask && li == 0) mask[(ob + ct * 32) >> 4] = (uint16_t)(bal >> (16 * kq));
    }
  }
}
static void rn_conv0_pool_reg_launch(const uint8_t* obs, const int32_t* idx, const float* W, const float* bias, float* pooled, uint8_t* pidx, int B,
                                     hipStream_t st, uint16_t* mask, float* pooled_relu = nullptr) {
  using G = RnPool0Reg;
  constexpr size_t lds = (size_t)G::NW * G::WAVE_FLOATS * sizeof(float);
  hipLaunchKernelGGL(rn_conv0_pool_reg_kernel, dim3((B + G::NW - 1) / G::NW), dim3(64 * G::NW), lds, st, obs, idx, W, bias, pooled, pidx, B, mask, pooled_relu);
}

// dgrad as a forward conv: Wt[(jh,jw)][co][ci] = W[(2-jh,2-jw)][ci][co]   (all 14 convs that need a dgrad, one launch)
struct RnFlipJob { int src, dst, ci, co; };
struct RnFlipJobs { RnFlipJob j[14]; int n; };
__global__ void rn_wflip_kernel(const float* P, float* Wt, RnFlipJobs jobs) {
  const RnFlipJob jb = jobs.j[blockIdx.y];
  const int n = 9 * jb.ci * jb.co;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int co = i % jb.co, t = i / jb.co, ci = t % jb.ci, tap = t / jb.ci;
    Wt[jb.dst + ((8 - tap) * jb.co + co) * jb.ci + ci] = P[jb.src + i];
  }
}

// ================================================================================================ weight gradient
// dW[(kh,kw,ci)][co] = sum over positions q of X[ci][q + kh*WP + kw] * dY[q][co]  — the same slab trick with the
// roles swapped: D rows = (tap, ci), D cols = co, MFMA reduction = positions.  Every wave keeps ALL tap tiles in
// registers and the 4 waves take interleaved position steps, so one dY fragment feeds 9 (5, 3) MFMAs.  Blocks are
// persistent over strips and emit one partial per block (fixed-order cross-wave sum), reduced by wgrad_reduce_kernel.
template <int CI_, int CO_, int H_, int R_, int NF_>
struct RnWGeom {
  using G = RnGeom<CI_, CO_, H_, R_, NF_>;
  static constexpr int CI = CI_, CO = CO_, H = H_, WP = G::WP, KK = G::KK;
  static constexpr bool M16 = G::M16;
  static constexpr int TPI = M16 ? 16 : 32;                       // D rows per tile
  static constexpr int NTI = (9 * CI + TPI - 1) / TPI;            // tap tiles per wave
  static constexpr int NKS = (G::NQ + KK - 1) / KK, KSW = (NKS + 3) / 4;  // position steps per strip / per wave
  static constexpr int NQP = KSW * 4 * KK;                        // padded positions
  static constexpr int S = NQP + 2 * WP + 3;
  static constexpr int PL = M16 ? rn_round_up_mod(S, 32, 2) : rn_round_up_mod(S, 32, 1);
  static constexpr int NPL = CI < 4 ? 4 : CI;
  static constexpr int SLAB = NPL * PL;
  static constexpr int DYS = NQP * CO;
  static constexpr int RED = 4 * TPI * CO;                        // cross-wave reduction scratch (one tile)
  static constexpr int LDS_FLOATS = SLAB + (DYS > RED ? DYS : RED);
  static constexpr int KX = 9 * CI;
  static constexpr bool PREFETCH = NF_ == 1 && NTI * (M16 ? 4 : 16) <= 96;   // next strip's loads in flight during the MFMA sweep (register budget)
  static int strips(int B) { return G::blocks(B); }
};

// POOLB: dY is not read but rebuilt from the gradient of the pooled map (dy = dout [B][H/2][H/2][CO]) and the pool's arg-max bytes —
// the max_pool backward of rn_pool_bwd_kernel (same (oh, ow) visiting order -> same bits) done while staging, so the un-pooled
// gradient (1.7 GB per 3840-frame minibatch for the first conv) is never written or read.  Used for the 84 -> 42 pool (pad_lo = 0).
template <class WG, bool U8, bool PRE_RELU, bool POOLB = false>
__global__ __launch_bounds__(256, 2) void rn_wgrad_kernel(const void* in_, const int32_t* idx, const float* dy, float* part, float* bpart, int B,
                                                       int nstrips, const uint8_t* pidx = nullptr) {
  using G = typename WG::G;
  constexpr int H = WG::H, CI = WG::CI, CO = WG::CO, WP = WG::WP, KK = WG::KK, NTI = WG::NTI, TPI = WG::TPI, PL = WG::PL;
  constexpr bool PF = WG::PREFETCH;
  static_assert(!POOLB || PF, "the pool-backward staging is wired into the prefetch path");
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  float* slab = rn_smem;
  float* dys = rn_smem + WG::SLAB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int v = tid; v < WG::LDS_FLOATS; v += 256) rn_smem[v] = 0.0f;   // pads / slack must be finite zeros for the whole kernel
  __syncthreads();

  // per-lane A bases: row i of tile T -> (tap, ci); invalid taps (k >= 9*CI) read tap 0 and are dropped at the end
  int abase[NTI];
  const int irow = WG::M16 ? (lane & 15) : (lane & 31), kpos = WG::M16 ? (lane >> 4) : (lane >> 5);
#pragma unroll
  for (int T = 0; T < NTI; ++T) {
    const int k = T * TPI + irow;
    int tap = k / (CI < 4 ? 4 : CI);
    const int ci = k - tap * (CI < 4 ? 4 : CI);
    if (tap > 8) tap = 0;
    abase[T] = ci * PL + (tap / 3) * WP + (tap % 3) + kpos;
  }
  const int ncol = WG::M16 ? (lane & 15) : (lane & 31);

  using AccT = typename std::conditional<WG::M16, rn_f32x4, rn_f32x16>::type;
  AccT acc[NTI];
#pragma unroll
  for (int T = 0; T < NTI; ++T)
#pragma unroll
    for (int e = 0; e < (WG::M16 ? 4 : 16); ++e) acc[T][e] = 0.0f;
  float bsum = 0.0f;

  // staged vectors per thread (strips of one frame are contiguous in memory: one base + per-item constant strides)
  constexpr int NVX = U8 ? G::SROWS * (H / 4) * 4 : G::SROWS * H * (CI < 4 ? 1 : CI / 4);
  constexpr int NVY = G::OROWS * H * (CO / 4);
  constexpr int NIX = (NVX + 255) / 256, NIY = (NVY + 255) / 256;
  float4 rx[PF && !U8 ? NIX : 1];
  uint32_t ru[PF && U8 ? NIX : 1];
  float4 ry[PF ? NIY : 1];

  auto where = [&](int st, int& b0, int& y0) {
    if (G::NF == 1) { b0 = st / G::STRIPS; y0 = (st - b0 * G::STRIPS) * G::R; }
    else { b0 = st * G::NF; y0 = 0; }
  };
  // gradient of max_pool(3,3) stride 2 SAME (pad_lo 0) at input pixel (y, x), channels 4g..4g+3 of frame f
  // gradient of max_pool(3,3) stride 2 SAME (pad_lo 0) at input pixel (y, x), channels 4g..4g+3 of frame f.  (A branch-free form with
  // four unconditional clamped loads measured slower: 1254 vs 1137 us for the fused kernel — most pixels have one or two windows.)
  auto pool_bwd_at = [&](int f, int y, int x, int g, bool ok) -> float4 {
    constexpr int HP = H / 2;
    float sx[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      for (int oh = (y - 1) / 2; 2 * oh <= y; ++oh) {   // windows covering row y, ascending (as rn_pool_bwd_kernel)
        if (oh < 0 || oh >= HP) continue;
        const int kh = y - 2 * oh;
        for (int ow = (x - 1) / 2; 2 * ow <= x; ++ow) {
          if (ow < 0 || ow >= HP) continue;
          const int kw = x - 2 * ow;
          const size_t o = (((size_t)(f * HP + oh) * HP + ow) * (CO / 4) + g) * 4;
          const uint32_t pi = *reinterpret_cast<const uint32_t*>(pidx + o);
          const float4 d = *reinterpret_cast<const float4*>(dy + o);
          const float e[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (((pi >> (8 * q)) & 0xffu) == (uint32_t)(kh * 3 + kw)) sx[q] += e[q];
        }
      }
    }
    return make_float4(sx[0], sx[1], sx[2], sx[3]);
  };
  auto fetch = [&](int b0, int y0) {
    if constexpr (U8) {
      const uint8_t* fr = (const uint8_t*)in_ + (size_t)(idx ? idx[b0] : b0) * CBM_FRAME;
#pragma unroll
      for (int it = 0; it < NIX; ++it) {
        const int v = min(tid + 256 * it, NVX - 1);
        const int cq = v % (H / 4), t = v / (H / 4), sr = t % G::SROWS, p = t / G::SROWS;
        const int y = y0 + sr - 1;
        ru[it] = *reinterpret_cast<const uint32_t*>(fr + (size_t)p * H * H + min(max(y, 0), H - 1) * H + 4 * cq);
      }
    } else {
      const float* in = (const float*)in_;
#pragma unroll
      for (int it = 0; it < NIX; ++it) {
        const int v = min(tid + 256 * it, NVX - 1);
        const int g = v % (CI / 4), pix = v / (CI / 4), c = pix % H, sr = pix / H;
        int f, y;
        if (G::NF == 1) { f = b0; y = y0 + sr - 1; }
        else { f = b0 + sr / (H + 1); y = sr % (H + 1) - 1; }
        rx[it] = *reinterpret_cast<const float4*>(in + ((size_t)(min(f, B - 1) * H + min(max(y, 0), H - 1)) * H + c) * CI + 4 * g);
      }
    }
#pragma unroll
    for (int it = 0; it < NIY; ++it) {
      const int v = min(tid + 256 * it, NVY - 1);
      const int g = v % (CO / 4), pix = v / (CO / 4), c = pix % H, orow = pix / H;
      int f, y;
      if (G::NF == 1) { f = b0; y = y0 + orow; }
      else { f = b0 + orow / (H + 1); y = orow % (H + 1); }
      if constexpr (!POOLB) ry[it] = *reinterpret_cast<const float4*>(dy + ((size_t)(min(f, B - 1) * H + min(y, H - 1)) * H + c) * CO + 4 * g);
    }
  };
  auto commit = [&](int b0, int y0) {
    if constexpr (U8) {
#pragma unroll
      for (int it = 0; it < NIX; ++it) {
        const int v = tid + 256 * it;
        if (NVX % 256 != 0 && v >= NVX) break;
        const int cq = v % (H / 4), t = v / (H / 4), sr = t % G::SROWS, p = t / G::SROWS;
        const int y = y0 + sr - 1;
        const bool ok = y >= 0 && y < H;
        float* d = slab + p * PL + 1 + sr * WP + 4 * cq;
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = ok ? cbm_u8_unit((uint8_t)((ru[it] >> (8 * q)) & 0xffu)) : 0.0f;
      }
    } else {
#pragma unroll
      for (int it = 0; it < NIX; ++it) {
        const int v = tid + 256 * it;
        if (NVX % 256 != 0 && v >= NVX) break;
        const int g = v % (CI / 4), pix = v / (CI / 4), c = pix % H, sr = pix / H;
        int f, y;
        if (G::NF == 1) { f = b0; y = y0 + sr - 1; }
        else { f = b0 + sr / (H + 1); y = sr % (H + 1) - 1; }
        const bool ok = y >= 0 && y < H && f < B;
        float* d = slab + (4 * g) * PL + 1 + sr * WP + c;
        const float e[4] = {rx[it].x, rx[it].y, rx[it].z, rx[it].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float val = ok ? e[q] : 0.0f;
          if (PRE_RELU) val = fmaxf(val, 0.0f);
          d[q * PL] = val;
        }
      }
    }
    // dY strip [position][co]; pad columns stay zero from the initial clear, rows outside the image are written as zeros
#pragma unroll
    for (int it = 0; it < NIY; ++it) {
      const int v = tid + 256 * it;
      if (NVY % 256 != 0 && v >= NVY) break;
      const int g = v % (CO / 4), pix = v / (CO / 4), c = pix % H, orow = pix / H;
      int f, y;
      if (G::NF == 1) { f = b0; y = y0 + orow; }
      else { f = b0 + orow / (H + 1); y = orow % (H + 1); }
      const bool ok = y < H && f < B;
      float4 x;
      if constexpr (POOLB) x = pool_bwd_at(f, y, c, g, ok);
      else x = ry[it];
      if (!ok) x = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(dys + (orow * WP + c) * CO + 4 * g) = x;
    }
  };

  int s = blockIdx.x, b0 = 0, y0 = 0;
  if constexpr (PF) { if (s < nstrips) { where(s, b0, y0); fetch(b0, y0); } }
  for (; s < nstrips; s += gridDim.x) {
    __syncthreads();  // previous strip fully consumed
    if constexpr (PF) {
      commit(b0, y0);
    } else {   // streaming copy (no registers held across the MFMA sweep: these geometries need them for accumulators)
      where(s, b0, y0);
    // ---- X slab (planes at pitch PL)
    if constexpr (U8) {
      constexpr int NV = G::SROWS * (H / 4) * 4;
      const int f = idx ? idx[b0] : b0;
      const uint8_t* fr = (const uint8_t*)in_ + (size_t)f * CBM_FRAME;
      for (int v = tid; v < NV; v += 256) {
        const int cq = v % (H / 4), t = v / (H / 4), sr = t % G::SROWS, p = t / G::SROWS;
        const int y = y0 + sr - 1;
        const bool ok = y >= 0 && y < H;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(fr + (size_t)p * H * H + min(max(y, 0), H - 1) * H + 4 * cq);
        float* d = slab + p * PL + 1 + sr * WP + 4 * cq;
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = ok ? cbm_u8_unit((uint8_t)((w >> (8 * q)) & 0xffu)) : 0.0f;
      }
    } else {
      constexpr int NV = G::SROWS * H * (CI / 4);
      const float* in = (const float*)in_;
      for (int v = tid; v < NV; v += 256) {
        const int g = v % (CI / 4), pix = v / (CI / 4), c = pix % H, sr = pix / H;
        int f, y;
        if (G::NF == 1) { f = b0; y = y0 + sr - 1; }
        else { f = b0 + sr / (H + 1); y = sr % (H + 1) - 1; }
        const bool ok = y >= 0 && y < H && f < B;
        const float4 x = *reinterpret_cast<const float4*>(in + ((size_t)(min(f, B - 1) * H + min(max(y, 0), H - 1)) * H + c) * CI + 4 * g);
        float* d = slab + (4 * g) * PL + 1 + sr * WP + c;
        float e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float val = ok ? e[q] : 0.0f;
          if (PRE_RELU) val = fmaxf(val, 0.0f);
          d[q * PL] = val;
        }
      }
    }
    // ---- dY strip [position][co], zero at pad columns / rows outside the image
    {
      constexpr int NV = G::OROWS * WP * (CO / 4);
      for (int v = tid; v < NV; v += 256) {
        const int g = v % (CO / 4), q = v / (CO / 4), orow = q / WP, c = q - orow * WP;
        int f, y;
        if (G::NF == 1) { f = b0; y = y0 + orow; }
        else { f = b0 + orow / (H + 1); y = orow % (H + 1); }
        const bool ok = c < H && y < H && f < B && (G::NF != 1 || orow < G::R);
        float4 x = *reinterpret_cast<const float4*>(dy + ((size_t)(min(f, B - 1) * H + min(y, H - 1)) * H + min(c, H - 1)) * CO + 4 * g);
        if (!ok) x = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(dys + q * CO + 4 * g) = x;
      }
    }
    }
    __syncthreads();
    if constexpr (PF) {
      const int sn = s + gridDim.x;
      if (sn < nstrips) { where(sn, b0, y0); fetch(b0, y0); }
    }
    // ---- bias partial: column sums of the strip (fixed order per thread)
    {
      constexpr int PARTS = 256 / CO;
      const int n = tid % CO, part_ = tid / CO;
      for (int q = part_; q < G::NQ; q += PARTS) bsum += dys[q * CO + n];
    }
    // ---- MFMA sweep: wave w takes position steps w, w+4, ...
    const float* bptr = dys + (wave * KK + kpos) * CO + ncol;
    const float* aptr = slab + wave * KK;
#pragma unroll 2
    for (int it = 0; it < WG::KSW; ++it) {
      const float bv = bptr[it * 4 * KK * CO];
      const float* ap = aptr + it * 4 * KK;
#pragma unroll
      for (int T = 0; T < NTI; ++T) {
        if constexpr (WG::M16) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[abase[T]], bv, acc[T], 0, 0, 0);
        else acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[abase[T]], bv, acc[T], 0, 0, 0);
      }
    }
  }

  // ---- one partial per block: sum the 4 waves' tiles in wave order through LDS
  float* red = dys;
  const int z = blockIdx.x;
#pragma unroll
  for (int T = 0; T < NTI; ++T) {
    __syncthreads();
    if constexpr (WG::M16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(wave * TPI + 4 * kpos + e) * CO + ncol] = acc[T][e];
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(wave * TPI + (e & 3) + 8 * (e >> 2) + 4 * kpos) * CO + ncol] = acc[T][e];
    }
    __syncthreads();
    for (int v = tid; v < TPI * CO; v += 256) {
      const int i = v / CO, n = v - i * CO, k = T * TPI + i;
      if (k < WG::KX) {
        float sum = red[v];
        for (int w = 1; w < 4; ++w) sum += red[w * TPI * CO + v];
        part[((size_t)z * WG::KX + k) * CO + n] = sum;
      }
    }
  }
  __syncthreads();
  red[tid] = bsum;
  __syncthreads();
  if (tid < CO) {
    float sum = red[tid];
    for (int q = 1; q < 256 / CO; ++q) sum += red[q * CO + tid];
    bpart[z * CO + tid] = sum;
  }
}

template <class WG, bool U8, bool PRE_RELU, bool POOLB = false>
static int rn_wgrad_launch(const void* in, const int32_t* idx, const float* dy, float* part, float* bpart, int B, int max_blocks, hipStream_t st,
                           const uint8_t* pidx = nullptr) {
  constexpr size_t lds = (size_t)WG::LDS_FLOATS * sizeof(float);
  static_assert(lds <= 160 * 1024, "wgrad slab exceeds LDS");
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)rn_wgrad_kernel<WG, U8, PRE_RELU, POOLB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int nstrips = WG::strips(B);
  const int nb = nstrips < max_blocks ? nstrips : max_blocks;
  hipLaunchKernelGGL((rn_wgrad_kernel<WG, U8, PRE_RELU, POOLB>), dim3(nb), dim3(256), lds, st, in, idx, dy, part, bpart, B, nstrips, pidx);
  return nb;
}

// ================================================================================================ weight gradient, load-unit fed (round 3)
// In dW[(tap,ci)][co] = sum_q X[q + tapoff][ci] * dY[q][co] BOTH MFMA operands have their lanes along a CHANNEL index (A: lane = ci of a tap tile,
// B: lane = co) and the reduction index is the position — so, unlike the forward / input-gradient convs (lanes = positions: channel planes), the
// weight gradient wants its slabs exactly as the activations lie in HBM (NHWC): X strip [(R+2) rows][WP pixels][CI] and dY strip [R rows][WP][CO],
// one zero pad pixel per row for the flat-index tap trick.  rn_wgrad_kernel above stages X as channel planes through registers (4 ds_write_b32 per
// float4 plus the index arithmetic, 256 VGPRs and spills for the 32-channel layers, no overlap with the sweep where the register budget is gone);
// here every image row of the strip is ONE linear copy by the load unit (global_load_lds_dwordx4: no staging registers, no VALU), strips are double
// buffered (the copy of strip s+1 runs under the sweep of strip s, one barrier per strip), one block per CU, NTI accumulator tiles per wave, the
// waves interleave position steps.  Same summation structure as rn_wgrad_kernel (per wave: its steps ascending over its strips ascending; then
// waves in order; then blocks in order by wgrad_reduce_kernel): deterministic, 1e-5 class.
template <int CI_, int CO_, int H_, int R_, int NF_, int NW_ = 4>
struct RnW2Geom {
  static constexpr int CI = CI_, CO = CO_, H = H_, R = R_, NF = NF_, WP = H + 1, NW = NW_;   // NW waves interleave the position steps
  static constexpr bool M16 = CO == 16;
  static constexpr int KK = M16 ? 4 : 2, TPI = M16 ? 16 : 32;
  static constexpr int OROWS = NF == 1 ? R : NF * (H + 1), SROWS = NF == 1 ? R + 2 : NF * (H + 1) + 1;
  static constexpr int NQ = OROWS * WP;
  static constexpr int NTI = (9 * CI + TPI - 1) / TPI;
  static constexpr int NKS = (NQ + KK - 1) / KK, KSW = (NKS + NW - 1) / NW, NQP = KSW * NW * KK;
  static constexpr int XPIX = NQP + 2 * WP + 3, XS = XPIX * CI, YS = NQP * CO;      // floats per stage
  static constexpr int STAGE = XS + YS;
  static constexpr int RED = NW * TPI * CO > NW * 64 ? NW * TPI * CO : NW * 64;
  static constexpr int LDS_FLOATS = 2 * STAGE > RED ? 2 * STAGE : RED;
  static constexpr int STRIPS = NF == 1 ? (H + R - 1) / R : 1;
  static constexpr int KX = 9 * CI;
  static int strips(int B) { return NF == 1 ? B * STRIPS : (B + NF - 1) / NF; }
};

// (hidden from the compiler, cbm_internal.h: through the builtin hipcc waited for the copy of strip s+1 in front of the sweep of strip s)
static __device__ __forceinline__ void rn_glds16(const void* g_lane, void* lds_wave_base) { cbm_glds16_hidden(g_lane, cbm_lds_addr(lds_wave_base)); }

template <class WG, bool RELU_PASS>
__global__ __launch_bounds__(WG::NW * 64, 1) void rn_wgrad2_kernel(const float* in, const float* dy, float* part, float* bpart, int B, int nstrips) {
  constexpr int H = WG::H, CI = WG::CI, CO = WG::CO, WP = WG::WP, KK = WG::KK, NTI = WG::NTI, TPI = WG::TPI, NW = WG::NW, NTHR = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float rn_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int v = tid; v < WG::LDS_FLOATS; v += NTHR) rn_smem[v] = 0.0f;   // pad pixels, gap rows and the slack behind a strip stay finite zeros
  __syncthreads();

  const int irow = WG::M16 ? (lane & 15) : (lane & 31), kpos = WG::M16 ? (lane >> 4) : (lane >> 5);
  // per-lane A offsets (floats, relative to position q's pixel): row irow of tap tile T -> (tap, ci); rows beyond 9*CI read tap 0 and are dropped
  int abase[NTI];
#pragma unroll
  for (int T = 0; T < NTI; ++T) {
    const int k = T * TPI + irow;
    int tap = k / CI;
    const int ci = k - tap * CI;
    if (tap > 8) tap = 0;
    abase[T] = ((tap / 3) * WP + (tap % 3) + kpos) * CI + ci;
  }
  using AccT = typename std::conditional<WG::M16, rn_f32x4, rn_f32x16>::type;
  AccT acc[NTI];
#pragma unroll
  for (int T = 0; T < NTI; ++T)
#pragma unroll
    for (int e = 0; e < (WG::M16 ? 4 : 16); ++e) acc[T][e] = 0.0f;
  float bsum = 0.0f;

  // one strip -> one stage: every image row is a contiguous run in HBM and (behind its pad pixel) in LDS; rows outside the image are zero-filled
  auto stage = [&](int st, int buf) __attribute__((always_inline)) {
    float* Xs = rn_smem + buf * WG::STAGE;
    float* Ys = Xs + WG::XS;
    int b0, y0;
    if (WG::NF == 1) { b0 = st / WG::STRIPS; y0 = (st - b0 * WG::STRIPS) * WG::R; } else { b0 = st * WG::NF; y0 = 0; }
    constexpr int PX = H * CI / 4, PY = H * CO / 4;                  // 16-byte pieces per image row
    constexpr int IX = (PX + 63) / 64, IY = (PY + 63) / 64;          // wave instructions per row
    // A row is one work item (dealt round robin over the waves) = IX / IY unrolled copies at constant offsets behind a wave-uniform row pointer:
    // dealing (row, 64-piece group) pairs cost a division and a 64-bit per-lane address per 1 KB copy
    // X rows: slab row sr holds image row y0 + sr - 1 (NF = 1) / row (sr % (H+1)) - 1 of frame b0 + sr / (H+1)
    for (int j = wave; j < WG::SROWS + WG::OROWS; j += NW) {
      if (j < WG::SROWS) {
        const int sr = j;
        int f, y;
        if (WG::NF == 1) { f = b0; y = y0 + sr - 1; } else { f = b0 + sr / (H + 1); y = sr % (H + 1) - 1; }
        const bool ok = y >= 0 && y < H && f < B;
        float* drow = Xs + (size_t)(1 + sr * WP) * CI;
        const float* src = in + ((size_t)(f * H + y) * H) * CI + lane * 4;
#pragma unroll
        for (int grp = 0; grp < IX; ++grp) {
          if (grp * 64 + 64 <= PX || lane < PX - grp * 64) {
            if (ok) rn_glds16(src + grp * 256, drow + grp * 256);
            else *reinterpret_cast<float4*>(drow + grp * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      } else {
        const int orow = j - WG::SROWS;
        int f, y;
        if (WG::NF == 1) { f = b0; y = y0 + orow; } else { f = b0 + orow / (H + 1); y = orow % (H + 1); }
        const bool ok = y < H && f < B && (WG::NF != 1 || orow < WG::R);
        float* drow = Ys + (size_t)(orow * WP) * CO;
        const float* src = dy + ((size_t)(f * H + y) * H) * CO + lane * 4;
#pragma unroll
        for (int grp = 0; grp < IY; ++grp) {
          if (grp * 64 + 64 <= PY || lane < PY - grp * 64) {
            if (ok) rn_glds16(src + grp * 256, drow + grp * 256);
            else *reinterpret_cast<float4*>(drow + grp * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
  };

  int s = blockIdx.x, n = 0;
  if (s < nstrips) stage(s, 0);
  for (; s < nstrips; s += gridDim.x, ++n) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();            // strip s has landed for every wave, and every wave is done with the other stage (strip s - gridDim.x)
    if (s + (int)gridDim.x < nstrips) stage(s + gridDim.x, (n + 1) & 1);
    float* Xs = rn_smem + (n & 1) * WG::STAGE;
    const float* Ys = Xs + WG::XS;
    if constexpr (RELU_PASS) {   // X is read through a relu (ResidualBlock's input): one pass over the landed strip instead of a v_max per A fragment
      for (int v = tid * 4; v < WG::XS; v += NTHR * 4) {
        float4 x = *reinterpret_cast<float4*>(Xs + v);
        x.x = fmaxf(x.x, 0.0f); x.y = fmaxf(x.y, 0.0f); x.z = fmaxf(x.z, 0.0f); x.w = fmaxf(x.w, 0.0f);
        *reinterpret_cast<float4*>(Xs + v) = x;
      }
      __syncthreads();
    }
    // wave w takes position steps w, w+NW, ...: step t covers positions q = t*KK + kpos.  One pointer per tap tile, every step a compile-time offset
    const float* bptr = Ys + (size_t)((wave * KK + kpos) * CO) + irow;
    const float* pT[NTI];
#pragma unroll
    for (int T = 0; T < NTI; ++T) pT[T] = Xs + (size_t)(wave * KK) * CI + abase[T];
    float fa[2][NTI], fb[2];
    auto frag = [&](int it, int set) __attribute__((always_inline)) {
      fb[set] = bptr[it * NW * KK * CO];
#pragma unroll
      for (int T = 0; T < NTI; ++T) fa[set][T] = pT[T][it * NW * KK * CI];
    };
    frag(0, 0);
#pragma unroll
    for (int it = 0; it < WG::KSW; ++it) {
      if (it + 1 < WG::KSW) frag(it + 1, (it + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const float bv = fb[it & 1];
      bsum += bv;
#pragma unroll
      for (int T = 0; T < NTI; ++T) {
        if constexpr (WG::M16) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[it & 1][T], bv, acc[T], 0, 0, 0);
        else acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[it & 1][T], bv, acc[T], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- one partial per block: sum the NW waves' tiles in wave order through LDS
  __syncthreads();
  float* red = rn_smem;
  const int z = blockIdx.x;
#pragma unroll
  for (int T = 0; T < NTI; ++T) {
    __syncthreads();
    if constexpr (WG::M16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(wave * TPI + 4 * kpos + e) * CO + irow] = acc[T][e];
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(wave * TPI + (e & 3) + 8 * (e >> 2) + 4 * kpos) * CO + irow] = acc[T][e];
    }
    __syncthreads();
    for (int v = tid; v < TPI * CO; v += NTHR) {
      const int i = v / CO, nn = v - i * CO, k = T * TPI + i;
      if (k < WG::KX) {
        float sum = red[v];
        for (int w = 1; w < NW; ++w) sum += red[w * TPI * CO + v];
        part[((size_t)z * WG::KX + k) * CO + nn] = sum;
      }
    }
  }
  // bias partial: lane (irow = co, kpos) of wave w holds the sum of dY[q][co] over its positions; fixed order: kpos ascending, then waves ascending
  __syncthreads();
  red[tid] = bsum;
  __syncthreads();
  if (tid < CO) {
    float sum = 0.0f;
    for (int w = 0; w < NW; ++w)
      for (int kp = 0; kp < KK; ++kp) sum += red[w * 64 + kp * TPI + tid];
    bpart[z * CO + tid] = sum;
  }
}

template <class WG, bool RELU_PASS>
static int rn_wgrad2_launch(const float* in, const float* dy, float* part, float* bpart, int B, int max_blocks, hipStream_t st) {
  constexpr size_t lds = (size_t)WG::LDS_FLOATS * sizeof(float);
  static_assert(lds <= 160 * 1024, "double-buffered wgrad strips exceed LDS");
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)rn_wgrad2_kernel<WG, RELU_PASS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  const int nstrips = WG::strips(B);
  const int per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : (int)(160 * 1024 / lds);
  int nb = 256 * per_cu;
  if (nb > max_blocks) nb = max_blocks;
  if (nb > nstrips) nb = nstrips;
  hipLaunchKernelGGL((rn_wgrad2_kernel<WG, RELU_PASS>), dim3(nb), dim3(WG::NW * 64), lds, st, in, dy, part, bpart, B, nstrips);
  return nb;
}

// ================================================================================================ first conv: weight gradient through the pool, sparse
// dW0[(kh,kw,c)][co] = sum over frames and conv positions of X[c][y+kh-1][x+kw-1] * dC0[y][x][co], and dC0 is the max-pool backward of the pooled
// gradient g: every pooled element (oh, ow, co) sends its gradient to ONE conv position, its window's arg-max.  rn_wgrad_kernel<.., POOLB> rebuilds
// the dense 84x84x16 dC0 strip by strip (up to four window look-ups per element) and multiplies all 7056 positions of a frame, three quarters of them
// zeros, on 16x16x4 MFMAs with a 36-row output (1.14-1.18 ms per 3840-frame minibatch, 27 TFLOP/s).  Here the sum runs over the POOLED elements:
//     dW0[(kh,kw,c)][co] += g[oh][ow][co] * X[c][ih+kh-1][iw+kw-1],   (ih, iw) = (2*oh + pkh, 2*ow + pkw) from the arg-max byte
// — a quarter of the multiply-adds, no pool-backward pass, no dense gradient.  The frame sits in LDS as BYTES with a zero border (pixels enter as exact
// integers, the 1/255 is applied once to the block's partial, as in conv1.hip); lane = (position slot, co) keeps its 36 + 1 sums in registers; a tap
// row's three pixels come from one ds_read2_b32 + v_alignbyte.  VALU work, not MFMA: 36 multiply-adds per pooled element is all there is.
#define RNS_PITCH 88                       // bytes per frame row in LDS: 4 zero bytes, 84 pixels (x = -1 reads byte 3, x = 84 the next row's byte 0)
#define RNS_PLANE (86 * RNS_PITCH)         // rows -1 .. 84
__global__ __launch_bounds__(256, 4) void rn_wgrad0_sparse_kernel(const uint8_t* obs, const int32_t* idx, const float* g, const uint8_t* pidx, float* part,
                                                                  float* bpart, int B) {
  __shared__ __attribute__((aligned(16))) unsigned char fb[4 * RNS_PLANE + 64];
  __shared__ float red[4][37][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, co = lane & 15, slot = lane >> 4;
  for (int v = tid; v < (4 * RNS_PLANE + 64) / 4; v += 256) reinterpret_cast<uint32_t*>(fb)[v] = 0u;   // borders stay zero for the whole kernel
  float acc[36], bacc = 0.0f;
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0f;
  // a frame is 7056 uint32 (4 pixels each), 28 per thread: uint32 v lies at byte 4v of the frame and goes to plane p = v / 1764, row r, column 4*cq
  constexpr int NV = 4 * 84 * 21, NI = (NV + 255) / 256;
  uint32_t ru[NI];
  auto fetch = [&](int f) {
    const uint32_t* fr = reinterpret_cast<const uint32_t*>(obs + (size_t)(idx ? idx[f] : f) * CBM_FRAME);
#pragma unroll
    for (int it = 0; it < NI; ++it) ru[it] = fr[min(tid + 256 * it, NV - 1)];
  };
  int f = blockIdx.x;
  if (f < B) fetch(f);
  for (; f < B; f += gridDim.x) {
    __syncthreads();            // everyone is done with the previous frame's bytes
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int v = tid + 256 * it;
      if (NV % 256 != 0 && v >= NV) break;
      const int rg = v / 21, p = v / 1764;          // global row p*84 + r
      // p * PLANE + (r + 1) * PITCH + 4 + 4 * cq with r = rg - 84 p, cq = v - 21 rg
      *reinterpret_cast<uint32_t*>(fb + 4 * v + (RNS_PITCH - 84) * rg + (RNS_PLANE - 84 * RNS_PITCH) * p + RNS_PITCH + 4) = ru[it];
    }
    __syncthreads();
    if (f + (int)gridDim.x < B) fetch(f + gridDim.x);
    // 1764 pooled positions, 16 per iteration over the block: wave w, slot s takes p = 16*it + 4*w + s
    const float* gf = g + (size_t)f * (1764 * 16) + co;
    const uint8_t* pf = pidx + (size_t)f * (1764 * 16) + co;
    int p = 4 * wave + slot, oh = 0, ow = p;          // p < 16 < 42
    float gn = gf[p * 16];
    uint32_t pn = pf[p * 16];
#pragma unroll 1
    for (int it = 0; it < 111; ++it) {
      const float gv = gn;
      const uint32_t pi = pn;
      const bool live = p < 1764;
      const int pnext = p + 16;
      if (pnext < 1764) { gn = gf[pnext * 16]; pn = pf[pnext * 16]; }
      if (live) {
        const int pkh = (int)((pi * 11u) >> 5), pkw = (int)pi - 3 * pkh;          // pi / 3, pi % 3 for pi in 0..8
        const int b0 = (2 * oh + pkh) * RNS_PITCH + 2 * ow + pkw + 3;               // byte of tap (0,0): pixel (ih-1, iw-1)
        const int sh = b0 & 3;
        const unsigned char* a0 = fb + (b0 & ~3);
        bacc += gv;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(a0 + c * RNS_PLANE + kh * RNS_PITCH);
            const uint32_t w = __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)sh);   // bytes b0 .. b0+3 of this row
            acc[(kh * 3 + 0) * 4 + c] = fmaf(gv, (float)(w & 255u), acc[(kh * 3 + 0) * 4 + c]);
            acc[(kh * 3 + 1) * 4 + c] = fmaf(gv, (float)((w >> 8) & 255u), acc[(kh * 3 + 1) * 4 + c]);
            acc[(kh * 3 + 2) * 4 + c] = fmaf(gv, (float)((w >> 16) & 255u), acc[(kh * 3 + 2) * 4 + c]);
          }
      }
      p = pnext; ow += 16;
      if (ow >= 42) { ow -= 42; oh += 1; }
    }
  }
  // block partial: slots of a wave (xor 16, 32), then the four waves in order
#pragma unroll
  for (int k = 0; k < 36; ++k) { acc[k] += __shfl_xor(acc[k], 16, 64); acc[k] += __shfl_xor(acc[k], 32, 64); }
  bacc += __shfl_xor(bacc, 16, 64); bacc += __shfl_xor(bacc, 32, 64);
  if (slot == 0) {
#pragma unroll
    for (int k = 0; k < 36; ++k) red[wave][k][co] = acc[k];
    red[wave][36][co] = bacc;
  }
  __syncthreads();
  const float inv255 = 1.0f / 255.0f;
  for (int v = tid; v < 37 * 16; v += 256) {
    const int k = v >> 4, n = v & 15;
    const float sum = ((red[0][k][n] + red[1][k][n]) + red[2][k][n]) + red[3][k][n];
    if (k < 36) part[((size_t)blockIdx.x * 36 + k) * 16 + n] = sum * inv255;
    else bpart[blockIdx.x * 16 + n] = sum;
  }
}
static int rn_wgrad0_sparse_launch(const uint8_t* obs, const int32_t* idx, const float* g, const uint8_t* pidx, float* part, float* bpart, int B, int max_blocks,
                                   hipStream_t st) {
  const int nb = B < max_blocks ? B : max_blocks;
  hipLaunchKernelGGL(rn_wgrad0_sparse_kernel, dim3(nb), dim3(256), 0, st, obs, idx, g, pidx, part, bpart, B);
  return nb;
}
