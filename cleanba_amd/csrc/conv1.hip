// conv1.hip — frame-resident kernels for the first convolution (8x8 stride 4, 4 -> 32 channels on uint8 frames).
//
// conv1 is the one GEMM of the network whose A operand is not fp32 in HBM: it is the im2col of the raw uint8
// NCHW frame (ppo:180-181 fold the NHWC transpose and the /255 into it).  Going through the generic gather
// (igemm.h) costs a 4-byte scattered load + index arithmetic + 4 conversions per 4 pixels, four times per pixel
// (each pixel sits in 2x2 patches) — measured staging-bound at 28-44 % of the f32 MFMA peak.  Here every frame is
// copied ONCE, coalesced (16-B loads), into LDS as bytes, and the MFMA A fragments are produced straight from
// those bytes (ds_read_u8 + exact /255) at the moment they are consumed; HBM sees each frame byte once per pass.
//
//   forward : block = persistent over frames, one frame of bytes in LDS at a time (400 positions = 12 tiles of 32 + a 16-row tail tile on
//             v_mfma_f32_16x16x4_f32); each wave owns 3 tiles and walks k = (c,kh,kw) ascending with all of them in flight (one B
//             fragment feeds 3 MFMAs).  Same fmaf chain as the oracle -> bit-exact.
//   wgrad   : block = persistent over a contiguous sample range; wave w owns k-tiles {2w, 2w+1} of dW[256][32];
//             dY rows are read straight from HBM/L2 (256 B per wave-load), frames from LDS.  Pixels enter as exact
//             integers and the 1/255 is applied once in the partial reduce (gradients carry a 1e-5 bar, not bits).
#include "cbm_internal.h"
#include "igemm.h"

#define FR 28224
// timing build only (-DCBM_CONV1_TRACE): shader-clock stamps of block 0's second frame, per wave and phase (tools/conv1_trace.py)
#ifdef CBM_CONV1_TRACE
__device__ unsigned long long cbm_conv1_trace[4][32];
extern "C" int cbm_debug_conv1_trace(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(cbm_conv1_trace), sizeof(cbm_conv1_trace)) == hipSuccess ? 0 : -1; }
#define C1T(k) do { if (blockIdx.x == 0 && s == s_lo + 1 && (threadIdx.x & 63) == 0) cbm_conv1_trace[threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C1T(k) do { } while (0)
#endif

// timing build only (-DCBM_BLOCK_TRACE): start / end wall-clock stamps (100 MHz, one clock for the whole chip) of EVERY block of the last launch of
// the two conv1 kernels (tools/block_trace.py: how unevenly do a persistent kernel's blocks finish beside the rollout?)
#ifdef CBM_BLOCK_TRACE
__device__ unsigned long long cbm_block_trace_c1[2][512][2];
extern "C" int cbm_debug_block_trace_conv1(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(cbm_block_trace_c1), sizeof(cbm_block_trace_c1)) == hipSuccess ? 0 : -1; }
#define BT_START(k) do { if (threadIdx.x == 0 && blockIdx.x < 512) cbm_block_trace_c1[k][blockIdx.x][0] = wall_clock64(); } while (0)
#define BT_END(k) do { if (threadIdx.x == 0 && blockIdx.x < 512) cbm_block_trace_c1[k][blockIdx.x][1] = wall_clock64(); } while (0)
#else
#define BT_START(k) do { } while (0)
#define BT_END(k) do { } while (0)
#endif

static __device__ __forceinline__ float relu_(float v) { return v > 0.0f ? v : 0.0f; }

// async global -> LDS copy of one 16-byte piece per lane (global_load_lds_dwordx4): the LDS destination is the
// wave-uniform base + lane*16, which is exactly a linear frame copy; no staging registers, vmcnt-tracked.
static __device__ __forceinline__ void glds16(const unsigned char* g_lane, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// copy one 28,224-byte frame (1764 pieces) into LDS with all 4 waves
static __device__ __forceinline__ void frame_to_lds(const uint8_t* frame, unsigned char* dst, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int v0 = (wave + 4 * j) * 64;  // wave-uniform first piece of this instruction
    if (v0 + lane < FR / 16) glds16(frame + (size_t)(v0 + lane) * 16, dst + (size_t)v0 * 16);
  }
}
// ------------------------------------------------------------------------------------------ forward
// Plane-streamed forward: a frame is 400 output positions = 12 tiles of 32 (three per wave; wave (w + f) % 4 takes tiles {first, first+4, first+8})
// + 16 left over, which are a 16x16x4 MFMA tile whose two 16-column halves go to two different waves (a fourth 32-row tile made one wave 33 % longer
// than the others at every barrier: 310 -> 265 us).  A first version built every A fragment from its byte at the moment of use — a pixel sits in
// 2x2 patches, so each was converted four times, 3-4 VALU instructions per MFMA, and VALU instructions take matrix-pipe issue slots on this chip
// (that kernel is in the git history, 306 us under load).  Here a frame goes through LDS one channel plane at a
// time AS FLOATS: the 7056 bytes of plane c are loaded to registers while plane c-1 is multiplied, converted once (1 VALU instruction per
// MFMA) and written as two half-planes [kw parity][84][42] — lane half h of a 32x32x2 MFMA supplies k = 2j + h, i.e. always one parity — so
// a tile's four kw-pairs of one (c,kh) patch row are 4 consecutive floats: two conflict-free ds_read_b64, no extraction, no conversion.
// LDS: 28,224 B plane + 32 KB weights, two blocks per CU; k = (c,kh,kw) ascending chain per output -> the oracle's bits.
__global__ __launch_bounds__(256, 2) void conv1_fwd_planes_kernel(const uint8_t* obs, const int32_t* idx, const float* W, const float* bias,
                                                                  float* out, uint32_t* mask, int S, int frames_per_block) {
  __shared__ __attribute__((aligned(16))) float smem_f[256 * 32 + 2 * 84 * 42];
  BT_START(0);
  float* Wl = smem_f;                 // [256][32]
  float* Pf = smem_f + 256 * 32;      // [2][84][42]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s_lo = blockIdx.x * frames_per_block, s_hi = min(S, s_lo + frames_per_block);
  uint32_t pw[7];
  auto load_plane = [&](const uint8_t* plane) __attribute__((always_inline)) {
    const uint32_t* g = reinterpret_cast<const uint32_t*>(plane);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int v = tid + 256 * j;
      pw[j] = g[min(v, 1763)];
    }
  };
  if (s_lo < s_hi) load_plane(obs + (size_t)(idx ? idx[s_lo] : s_lo) * FR);
  for (int i = tid; i < 256 * 32 / 4; i += 256) {
    const int k = i >> 3, n4 = (i & 7) * 4;
    const int c = k >> 6, kh = (k >> 3) & 7, kw = k & 7;
    *reinterpret_cast<float4*>(Wl + k * 32 + n4) = *reinterpret_cast<const float4*>(W + ((kh * 8 + kw) * 4 + c) * 32 + n4);
  }
  float bch[16];                                            // bias of this lane's 16 channels: c(e) = (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
  for (int e = 0; e < 16; ++e) bch[e] = bias[(e & 3) + 8 * (e >> 2) + 4 * h];
  constexpr int MAXT = 3;
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const int r16 = lane & 15, g4 = lane >> 4;
  for (int s = s_lo; s < s_hi; ++s) {
    const int first = (wave + (s - s_lo)) & 3;
    const bool four = first < 2;
    const int tj = first & 1;
    int base[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int p = min((first + 4 * t) * 32 + li, 399);
      const int oh = p / 20, ow = p - oh * 20;
      base[t] = h * 3528 + oh * 4 * 42 + ow * 2;
    }
    // tail tile (positions 384..399 = row 19, columns 4..19): lane (g4, r16) supplies kw = 4*st + g4 -> parity g4 & 1, half-plane column 2*ow + (g4 >> 1) + 2*st
    const int tbase = (g4 & 1) * 3528 + ((384 + r16) / 20) * 4 * 42 + ((384 + r16) % 20) * 2 + (g4 >> 1);
    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    f32x4_t tacc = {0.f, 0.f, 0.f, 0.f};
    const uint8_t* frame = obs + (size_t)(idx ? idx[s] : s) * FR;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      C1T(5 * c);
      __syncthreads();                       // every wave is done with the previous plane (and, for c == 0, with the weights' staging loads)
      C1T(5 * c + 1);
#ifndef C1F_ABL   // timing builds only: 1 no epilogue stores, 2 planes converted / stored to LDS only for the block's first plane, 4 no plane loads after the first, 16 one barrier per plane
#define C1F_ABL 0
#endif
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int v = tid + 256 * j;
        if ((C1F_ABL & 2) && !(s == s_lo && c == 0)) break;
        if (v < 1764) {
          const int y = v / 21, xw = v - y * 21;
          const uint32_t w = pw[j];
          f32x2_t ev = {cbm_u8_unit(w & 255u), cbm_u8_unit((w >> 16) & 255u)}, od = {cbm_u8_unit((w >> 8) & 255u), cbm_u8_unit(w >> 24)};
          *reinterpret_cast<f32x2_t*>(Pf + y * 42 + 2 * xw) = ev;
          *reinterpret_cast<f32x2_t*>(Pf + 3528 + y * 42 + 2 * xw) = od;
        }
      }
      if (!(C1F_ABL & 4)) {
      if (c < 3) load_plane(frame + (c + 1) * 7056);                                           // lands while this plane is multiplied
      else if (s + 1 < s_hi) load_plane(obs + (size_t)(idx ? idx[s + 1] : s + 1) * FR);
      }
      C1T(5 * c + 2);
      if (!(C1F_ABL & 16)) __syncthreads();
      C1T(5 * c + 3);
      f32x2_t aa[2][MAXT][2];
      float ta[2][2], wv[2][6];
      auto fetch = [&](int kh, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          const float* q = Pf + base[t] + kh * 42;
          aa[set][t][0] = *reinterpret_cast<const f32x2_t*>(q);
          aa[set][t][1] = *reinterpret_cast<const f32x2_t*>(q + 2);
        }
        const int krow = (c * 8 + kh) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[set][j] = Wl[(krow + 2 * j + h) * 32 + li];
        if (four) {
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            ta[set][st] = Pf[tbase + kh * 42 + 2 * st];
            wv[set][4 + st] = Wl[(krow + 4 * st + g4) * 32 + 16 * tj + r16];
          }
        }
      };
      auto fma_row = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < MAXT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[set][j], aa[set][t][j >> 1][j & 1], acc[t], 0, 0, 0);   // D[channel][position]: see the epilogue
        if (four) {
#pragma unroll
          for (int st = 0; st < 2; ++st) tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[set][4 + st], ta[set][st], tacc, 0, 0, 0);
        }
      };
      fetch(0, 0);
#pragma unroll
      for (int kh = 0; kh < 8; ++kh) {
        if (kh + 1 < 8) fetch(kh + 1, (kh + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        fma_row(kh & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      C1T(5 * c + 4);
    }
    C1T(20);
#ifdef CBM_CONV1_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    C1T(22);
#endif
    if (C1F_ABL & 1) {
      float sabl = tacc[0];
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) sabl += acc[t][e];
      if (sabl == 1.2345e-33f) out[0] = sabl;
      continue;
    }
    // The products are formed TRANSPOSED — the weights are the MFMA's A operand, the pixels its B operand (the same two registers, swapped: same products,
    // same k order, same bits) — so that a lane ends up with 16 CHANNELS of ONE position, in four runs of four: four 16-byte stores per tile instead of
    // sixteen 4-byte ones, and the position's ReLU word is this lane's own bits | its partner half's (one cross-half exchange) instead of sixteen
    // ballots handed round with two v_writelane each.  (Round 6: with NO epilogue the kernel is 18 us shorter, 254 -> 236, profiles/r06_pf2_ablation.txt — and
    // this four-times-leaner one measures the SAME 251-256 us as the old: what the epilogue costs is its 197 MB of writes, not its instructions.  Kept: less code.)
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      C1T(23 + t);
      const int m = (first + 4 * t) * 32 + li;                   // position (tiles 0..11: always < 384)
      float* o = out + ((size_t)s * 400 + m) * 32 + 4 * h;
      uint32_t bits = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = relu_(acc[t][4 * g] + bch[4 * g]); v.y = relu_(acc[t][4 * g + 1] + bch[4 * g + 1]);
        v.z = relu_(acc[t][4 * g + 2] + bch[4 * g + 2]); v.w = relu_(acc[t][4 * g + 3] + bch[4 * g + 3]);
        *reinterpret_cast<float4*>(o + 8 * g) = v;               // channels 8g + 4h .. + 3
        bits |= ((v.x > 0.0f ? 1u : 0u) | (v.y > 0.0f ? 2u : 0u) | (v.z > 0.0f ? 4u : 0u) | (v.w > 0.0f ? 8u : 0u)) << (8 * g + 4 * h);
      }
      bits |= (uint32_t)__shfl_xor((int)bits, 32, 64);
      if (mask && lane < 32) mask[(size_t)s * 400 + m] = bits;
    }
    C1T(26);
    if (four) {   // tail tile, also transposed: lane (g4, r16) holds channels 16 tj + 4 g4 .. + 3 of position 384 + r16
      const float4 bt = *reinterpret_cast<const float4*>(bias + 16 * tj + 4 * g4);
      float4 v;
      v.x = relu_(tacc[0] + bt.x); v.y = relu_(tacc[1] + bt.y); v.z = relu_(tacc[2] + bt.z); v.w = relu_(tacc[3] + bt.w);
      *reinterpret_cast<float4*>(out + ((size_t)s * 400 + 384 + r16) * 32 + 16 * tj + 4 * g4) = v;
      uint32_t bits = ((v.x > 0.0f ? 1u : 0u) | (v.y > 0.0f ? 2u : 0u) | (v.z > 0.0f ? 4u : 0u) | (v.w > 0.0f ? 8u : 0u)) << (4 * g4);
      bits |= (uint32_t)__shfl_xor((int)bits, 16, 64);
      bits |= (uint32_t)__shfl_xor((int)bits, 32, 64);
      if (mask && lane < 16) reinterpret_cast<uint16_t*>(mask + (size_t)s * 400 + 384 + lane)[tj] = (uint16_t)bits;
    }
    C1T(21);
  }
  BT_END(0);
}

// ------------------------------------------------------------------------------------------ forward, exact products on the bf16 matrix cores
// conv1's A operand is special: a pixel is an integer 0..255, which a bf16 (8 significant bits) holds EXACTLY.  So only the weight needs splitting:
// w/255 (fp32) = w1 + w2 + w3 with each term the next 8 significant bits (truncation, so every remainder is exact and the third term holds what is
// left: 24 = 8 + 8 + 8), and px * w/255 = px*w1 + px*w2 + px*w3 where every product is EXACT (8 x 8 bits) and the sums are fp32 — no bit of either
// operand is dropped; what differs from the fp32 fmaf chain is the order of the roundings only (measured against the oracle's chain:
// tests/test_gpu_conv1_exact.py).  v_mfma_f32_32x32x16_bf16 retires sixteen k per 8 passes where v_mfma_f32_32x32x2_f32 retires two per 16: three
// products cost 3/16 of the fp32 kernel's matrix time, and the kernel becomes what the layer's bytes say it should be — HBM-bound (108 MB of frames
// in, 197 MB of activations + 6 MB of ReLU words out).
//   LDS: the three weight terms as MFMA A fragments  Wl[term][q][h][channel][8 kw]  (48 KB, 16 bytes per lane: conflict-free ds_read_b128)
//        + ONE frame as bytes (28,224 B); two blocks per CU cover each other's frame turn-over, the next frame's bytes wait in registers.
//   K step q = (c, kh pair): lane (position, h) needs the 8 bytes of patch row kh = 2 (q & 3) + h — contiguous in the frame — converted with
//        v_cvt_f32_ubyte + v_perm (12 VALU per fragment, used by three MFMAs); k order inside an instruction is (kh parity, kw).
//   13 tiles of 32 positions per frame (the last one half empty: the matrix time is not what bounds this kernel), tiles {first, first+4, first+8[, 12]}
//        per wave with `first` rotating from frame to frame; products formed transposed (weights = the MFMA's A operand) so that the epilogue is the fp32
//        kernel's: a lane holds 16 channels of one position -> four 16-byte stores and the position's ReLU word.
typedef uint32_t c1_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t c1_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 c1_bf16x8 __attribute__((ext_vector_type(8)));
static __device__ __forceinline__ uint32_t c1_pack_hi16(float lo, float hi) {   // the two upper halves (bf16 by truncation; exact for integers < 256)
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
static __device__ __forceinline__ c1_bf16x8 c1_px8_bf16(uint32_t w0, uint32_t w1) {
  c1_u32x4 r;
  r[0] = c1_pack_hi16((float)(w0 & 255u), (float)((w0 >> 8) & 255u));
  r[1] = c1_pack_hi16((float)((w0 >> 16) & 255u), (float)(w0 >> 24));
  r[2] = c1_pack_hi16((float)(w1 & 255u), (float)((w1 >> 8) & 255u));
  r[3] = c1_pack_hi16((float)((w1 >> 16) & 255u), (float)(w1 >> 24));
  return __builtin_bit_cast(c1_bf16x8, r);
}
// three bf16 terms of an fp32 value by truncation: t1 + t2 + t3 == v exactly (for |v| >= 2^-103 or 0: below that the last remainder is subnormal and its
// upper 16 bits drop less than 2^-133; tests/test_conv1_exact_split.py restates this in numpy)
static __device__ __forceinline__ void c1_split3(float v, uint32_t& t1, uint32_t& t2, uint32_t& t3) {
  t1 = __float_as_uint(v) & 0xffff0000u;
  const float r1 = v - __uint_as_float(t1);
  t2 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(t2);
  t3 = __float_as_uint(r2);            // at most 8 significant bits left: its lower half is zero
}
#define C1X_WBYTES (3 * 16 * 2 * 32 * 16)
#ifndef C1X_ABL   // timing builds only (results wrong on purpose): 1 frame turn-over in front of the epilogue stores, 2 no epilogue, 4 no byte -> bf16 conversion,
#define C1X_ABL 0 //   8 one weight fragment set for all K steps (still read each step: same address), 16 no MFMAs
#endif
#ifndef C1W_ABL   // timing builds only: 1 dY one group ahead, 2 no split of dY, 4 no A-fragment gathers, 8 no MFMAs, 16 no closing reduction, 32 frame converted once per block
#define C1W_ABL 0
#endif
template <bool MASK>
__global__ __launch_bounds__(256, 2) void conv1_fwd_exact_kernel(const uint8_t* obs, const int32_t* idx, const float* W, const float* bias,
                                                                 float* out, uint32_t* mask, int S, int frames_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char smem_x[C1X_WBYTES + FR + 16];
  c1_u32x4* Wl = reinterpret_cast<c1_u32x4*>(smem_x);   // [term][q][h][channel] x 8 bf16
  unsigned char* F = smem_x + C1X_WBYTES;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s_lo = blockIdx.x * frames_per_block, s_hi = min(S, s_lo + frames_per_block);
  if (s_lo >= s_hi) return;
  c1_u32x4 pw[7];
  auto load_frame = [&](const uint8_t* frame) __attribute__((always_inline)) {
    const c1_u32x4* g = reinterpret_cast<const c1_u32x4*>(frame);
#pragma unroll
    for (int j = 0; j < 7; ++j) pw[j] = g[min(tid + 256 * j, FR / 16 - 1)];
  };
  auto put_frame = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int v = tid + 256 * j;
      if (v < FR / 16) *reinterpret_cast<c1_u32x4*>(F + 16 * v) = pw[j];
    }
  };
  load_frame(obs + (size_t)(idx ? idx[s_lo] : s_lo) * FR);
  // weight terms: item = (q, h, channel) -> the 8 kw of patch row (c = q >> 2, kh = 2 (q & 3) + h)
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    const int item = tid + 256 * it, n = item & 31, hh = (item >> 5) & 1, q = item >> 6;
    const int c = q >> 2, kh = 2 * (q & 3) + hh;
    uint32_t t1[8], t2[8], t3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c1_split3(W[((kh * 8 + j) * 4 + c) * 32 + n] / 255.0f, t1[j], t2[j], t3[j]);
    c1_u32x4 v1, v2, v3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v1[j] = (t1[2 * j] >> 16) | t1[2 * j + 1];
      v2[j] = (t2[2 * j] >> 16) | t2[2 * j + 1];
      v3[j] = (t3[2 * j] >> 16) | (t3[2 * j + 1] & 0xffff0000u);
    }
    Wl[((0 * 16 + q) * 2 + hh) * 32 + n] = v1;
    Wl[((1 * 16 + q) * 2 + hh) * 32 + n] = v2;
    Wl[((2 * 16 + q) * 2 + hh) * 32 + n] = v3;
  }
  float bch[16];                                            // bias of this lane's 16 channels: c(e) = (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
  for (int e = 0; e < 16; ++e) bch[e] = bias[(e & 3) + 8 * (e >> 2) + 4 * h];
  put_frame();
  if (s_lo + 1 < s_hi) load_frame(obs + (size_t)(idx ? idx[s_lo + 1] : s_lo + 1) * FR);
  __syncthreads();
  // Per frame: multiply (LDS = frame s) + epilogue stores | barrier | frame s+1: registers -> LDS, then the loads of frame s+2 | barrier.
  for (int s = s_lo; s < s_hi; ++s) {
    const int first = (wave + (s - s_lo)) & 3;
    const bool four = first == 0;
    int base[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int p = min((first + 4 * t) * 32 + li, 399);
      const int oh = p / 20, ow = p - oh * 20;
      base[t] = (oh * 4 + h) * 84 + ow * 4;
    }
    auto frame_tiles = [&](auto nt_) __attribute__((always_inline)) {
      constexpr int NT = decltype(nt_)::value;
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int foff = (q >> 2) * 7056 + (q & 3) * 168;
        c1_bf16x8 wf[3], xb[NT];
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) wf[tm] = __builtin_bit_cast(c1_bf16x8, Wl[((tm * 16 + ((C1X_ABL & 8) ? 0 : q)) * 2 + h) * 32 + li]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const uint32_t* px = reinterpret_cast<const uint32_t*>(F + base[t] + foff);
          if (C1X_ABL & 4) { c1_u32x4 r = {px[0], px[1], px[0], px[1]}; xb[t] = __builtin_bit_cast(c1_bf16x8, r); }
          else xb[t] = c1_px8_bf16(px[0], px[1]);
        }
#pragma unroll
        for (int tm = 2; tm >= 0; --tm)          // small terms first; consecutive MFMAs go to different accumulators
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (C1X_ABL & 16) { acc[t][tm] += (float)xb[t][tm] * (float)wf[tm][t]; continue; }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tm], xb[t], acc[t], 0, 0, 0);
          }
      }
      if (C1X_ABL & 1) {               // timing build: frame turn-over in FRONT of the epilogue stores (the LDS writes then never wait behind this frame's stores
        __syncthreads();               // on the in-order vmcnt; measured 14 us SLOWER per 3840 frames: the accumulators sit through a barrier and the stores start later)
        if (s + 1 < s_hi) put_frame();
        if (s + 2 < s_hi) load_frame(obs + (size_t)(idx ? idx[s + 2] : s + 2) * FR);
      }
      if (C1X_ABL & 2) {
        float sabl = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) sabl += acc[t][e];
        if (sabl == 1.2345e-33f) out[0] = sabl;
        return;
      }
      // epilogue: the fp32 kernel's (D[channel][position]: a lane holds 16 channels of one position)
#pragma unroll
      // No branch around the stores: the lanes of the half-empty 13th tile computed position 399 (base[] clamps) and store it again — same address, same
      // bits — and both lane halves store the position's ReLU word.  With a branch hipcc cannot count the stores in flight and waits for ALL of them
      // (vmcnt(0)) in front of the next frame's LDS writes, whose loads were issued a whole frame earlier.
      for (int t = 0; t < NT; ++t) {
        const int m = min((first + 4 * t) * 32 + li, 399);
        float* o = out + ((size_t)s * 400 + m) * 32 + 4 * h;
        uint32_t bits = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v;
          v.x = relu_(acc[t][4 * g] + bch[4 * g]); v.y = relu_(acc[t][4 * g + 1] + bch[4 * g + 1]);
          v.z = relu_(acc[t][4 * g + 2] + bch[4 * g + 2]); v.w = relu_(acc[t][4 * g + 3] + bch[4 * g + 3]);
          *reinterpret_cast<float4*>(o + 8 * g) = v;               // channels 8g + 4h .. + 3
          bits |= ((v.x > 0.0f ? 1u : 0u) | (v.y > 0.0f ? 2u : 0u) | (v.z > 0.0f ? 4u : 0u) | (v.w > 0.0f ? 8u : 0u)) << (8 * g + 4 * h);
        }
        bits |= (uint32_t)__shfl_xor((int)bits, 32, 64);
        if (MASK) mask[(size_t)s * 400 + m] = bits;
      }
    };
    if (four) frame_tiles(std::integral_constant<int, 4>{});
    else frame_tiles(std::integral_constant<int, 3>{});
    if (!(C1X_ABL & 1)) {
      __syncthreads();                 // every wave is done with this frame's bytes
      if (s + 1 < s_hi) put_frame();
      if (s + 2 < s_hi) load_frame(obs + (size_t)(idx ? idx[s + 2] : s + 2) * FR);   // two frames ahead: lands while the next frame is multiplied
    }
    __syncthreads();
  }
}

#ifndef CBM_C1X_BLOCKS
#define CBM_C1X_BLOCKS 1024
#endif
static const int C1X_BLOCKS = [] { const char* e = getenv("CBM_C1X_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : CBM_C1X_BLOCKS; }();
void launch_conv1_fwd_frames(const uint8_t* obs, const int32_t* idx, const float* W, const float* bias, float* out, uint32_t* mask, int S,
                             hipStream_t st, bool exact) {
  // two frames per block at 3840 frames.  512 persistent blocks (exactly two per CU) were fragile under the concurrent rollout: a CU that could
  // not take its second block (LDS held by actor blocks) left a straggler — 375 us under load against 286 isolated; with many short blocks the
  // dispatcher balances.  Under load / isolated: 4096 blocks 279 / 260 us, 2048 275 / 253, 1280 283 / 276, 768 292 / 245
  int blocks = exact ? C1X_BLOCKS : 2048;
  if (S < blocks) blocks = S;
  const int fpb = (S + blocks - 1) / blocks;
  blocks = (S + fpb - 1) / fpb;
  if (exact && mask) hipLaunchKernelGGL(conv1_fwd_exact_kernel<true>, dim3(blocks), dim3(256), 0, st, obs, idx, W, bias, out, mask, S, fpb);
  else if (exact) hipLaunchKernelGGL(conv1_fwd_exact_kernel<false>, dim3(blocks), dim3(256), 0, st, obs, idx, W, bias, out, mask, S, fpb);
  else hipLaunchKernelGGL(conv1_fwd_planes_kernel, dim3(blocks), dim3(256), 0, st, obs, idx, W, bias, out, mask, S, fpb);
}

// ------------------------------------------------------------------------------------------ wgrad
// part[z][k=(c,kh,kw)][n] (pixel units: the reduce multiplies by 1/255), bpart[z][n].
// One frame of bytes in LDS (28 KB): 4-5 blocks per CU overlap each other's staging.
__global__ __launch_bounds__(256, 2) void conv1_wgrad_frames_kernel(const uint8_t* obs, const int32_t* idx, const float* dy, float* part,
                                                                    float* bpart, int S, int frames_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char F[FR];
  BT_START(1);
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s_lo = blockIdx.x * frames_per_block, s_hi = min(S, s_lo + frames_per_block);
  // this wave's two k-tiles: k = 64*wave + 32*t + li  ->  byte offset of (c,kh,kw) inside the frame
  int kb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = 64 * wave + 32 * t + li;
    kb[t] = (k >> 6) * 7056 + ((k >> 3) & 7) * 84 + (k & 7) + 4 * h;  // + position offset of the m-pair member h
  }
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
  float bs = 0.0f;
  for (int s = s_lo; s < s_hi; ++s) {
    __syncthreads();  // previous frame fully consumed
    const int f = idx ? idx[s] : s;
    frame_to_lds(obs + (size_t)f * FR, F, wave, lane);
    const float* g = dy + (size_t)s * 400 * 32 + h * 32 + li;  // row m0+h, column li
    float bc[10], bn[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) bc[q] = g[(size_t)(2 * q) * 32];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int oh = 0; oh < 20; ++oh) {
      if (oh + 1 < 20) {
#pragma unroll
        for (int q = 0; q < 10; ++q) bn[q] = g[(size_t)((oh + 1) * 20 + 2 * q) * 32];
      }
      const int rowoff = oh * 4 * 84;
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        const int mo = rowoff + 8 * q;  // ow = 2q (+1 for h=1 folded into kb)
        const float b = bc[q];
        bs += b;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float a = (float)F[kb[t] + mo];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 10; ++q) bc[q] = bn[q];
    }
  }
  // partial out: rows k = 64*wave + 32*t + row, column li
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      part[((size_t)blockIdx.x * 256 + 64 * wave + 32 * t + row) * 32 + li] = acc[t][e];
    }
  if (wave == 0) {
    bs += __shfl_xor(bs, 32, 64);
    if (h == 0) bpart[blockIdx.x * 32 + li] = bs;
  }
  BT_END(1);
}

// Split-bf16 flavour (cbm_config.backward_split): pixels are integers 0..255, exact in bf16, so only dY needs splitting (two terms):
// dW += px * dy1 + px * dy2 on v_mfma_f32_32x32x16_bf16, sixteen output positions per instruction instead of two.  Same frame-resident
// structure, same partial layout (the reduce applies 1/255).  A frame's 400 positions are 25 groups of 16 consecutive positions; lane
// (li, h) supplies positions 8h..8h+7 of the group for k-row li (A) / channel li (B).
__global__ __launch_bounds__(256, 2) void conv1_wgrad_frames_split_kernel(const uint8_t* obs, const int32_t* idx, const float* dy, float* part,
                                                                          float* bpart, int S, int frames_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char F[FR];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s_lo = blockIdx.x * frames_per_block, s_hi = min(S, s_lo + frames_per_block);
  int kb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = 64 * wave + 32 * t + li;
    kb[t] = (k >> 6) * 7056 + ((k >> 3) & 7) * 84 + (k & 7);
  }
  f32x16 acc[2], lo[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[t][e] = 0.0f; lo[t][e] = 0.0f; }
  float bs = 0.0f;
  for (int s = s_lo; s < s_hi; ++s) {
    __syncthreads();  // previous frame fully consumed
    const int f = idx ? idx[s] : s;
    frame_to_lds(obs + (size_t)f * FR, F, wave, lane);
    const float* g = dy + ((size_t)s * 400 + 8 * h) * 32 + li;  // position 8h of group 0, channel li
    float bc[8], bn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bc[j] = g[j * 32];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int grp = 0; grp < 25; ++grp) {
      if (grp + 1 < 25) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bn[j] = g[((grp + 1) * 16 + j) * 32];
      }
      const int p0 = grp * 16 + 8 * h, oh0 = p0 / 20, ow0 = p0 - oh0 * 20;
      c1_bf16x8 b1, b2, a[2];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = bc[j];
        bs += v;
        const __bf16 t1 = (__bf16)v;
        b1[j] = t1;
        b2[j] = (__bf16)(v - (float)t1);
        const int off = oh0 * 336 + (ow0 + j) * 4 + (ow0 + j >= 20 ? 336 - 80 : 0);   // position p0 + j -> byte offset of its patch origin
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t][j] = (__bf16)(float)F[kb[t] + off];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b1, acc[t], 0, 0, 0);
        lo[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b2, lo[t], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) bc[j] = bn[j];
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      part[((size_t)blockIdx.x * 256 + 64 * wave + 32 * t + row) * 32 + li] = acc[t][e] + lo[t][e];
    }
  if (wave == 0) {
    bs += __shfl_xor(bs, 32, 64);
    if (h == 0) bpart[blockIdx.x * 32 + li] = bs;
  }
}

// Exact-product flavour (the default at learner sizes, cbm_config.conv1_fp32_chain = 0): like the forward above, the pixel operand is exact in bf16,
// so dW = sum_pos px * dY needs only dY split — into THREE terms that sum to it exactly (c1_split3) — and every product px * dy_i is exact; the sums are
// fp32.  v_mfma_f32_32x32x16_bf16 takes sixteen positions per instruction.  The frame is converted ONCE per block and frame into bf16 (LDS), so an A
// fragment is eight 16-bit LDS reads and four packs, no per-use conversion.  Work split: a frame's 400 positions are 25 groups of 16; a wave takes every
// fourth group and ALL eight k-tiles of dW[256][32] for it (128 accumulator registers), so the split of dY (5.5 VALU instructions per element) and
// the row-wrap address arithmetic are paid once per 24 MFMAs; the four waves' sums are added through LDS when the block ends (two rounds).
// Partials part[z][k][n] in pixel units, as the fp32 kernel writes them (the reduce applies 1/255).
// This is the FIRST form (row-major bf16 frame in LDS, eight 16-bit gathers per fragment: 122-130 us per 3840 frames), kept for A/B runs (CBM_C1W_DL=0);
// what ships is conv1_wgrad_exact_kernel below (de-interleaved LDS frame, 105-110 us).
#define C1WX_LDS 65536
__global__ __launch_bounds__(256, 2) void conv1_wgrad_exact_rowmajor_kernel(const uint8_t* obs, const int32_t* idx, const float* dy, float* part,
                                                                   float* bpart, int S, int frames_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned short FB[C1WX_LDS / 2];   // one frame as bf16 (56,448 B); 64 KB for the closing reduction
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s_lo = blockIdx.x * frames_per_block, s_hi = min(S, s_lo + frames_per_block);
  if (s_lo >= s_hi) return;
  const int kb = (li >> 3) * 84 + (li & 7);          // tap (kh' = li >> 3, kw = li & 7) of k-tile tt = (c = tt >> 1, kh = 4 (tt & 1) + kh'): + c * 7056 + (tt & 1) * 336
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
  float bs = 0.0f;
  c1_u32x4 pw[7];
  auto load_frame = [&](const uint8_t* frame) __attribute__((always_inline)) {
    const c1_u32x4* g = reinterpret_cast<const c1_u32x4*>(frame);
#pragma unroll
    for (int j = 0; j < 7; ++j) pw[j] = g[min(tid + 256 * j, FR / 16 - 1)];
  };
  load_frame(obs + (size_t)(idx ? idx[s_lo] : s_lo) * FR);
  for (int s = s_lo; s < s_hi; ++s) {
    __syncthreads();  // previous frame fully consumed
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int v = tid + 256 * j;
      if ((C1W_ABL & 32) && s != s_lo) break;
      if (v < FR / 16) {
        *reinterpret_cast<c1_u32x4*>(FB + 16 * v) = __builtin_bit_cast(c1_u32x4, c1_px8_bf16(pw[j][0], pw[j][1]));
        *reinterpret_cast<c1_u32x4*>(FB + 16 * v + 8) = __builtin_bit_cast(c1_u32x4, c1_px8_bf16(pw[j][2], pw[j][3]));
      }
    }
    if (s + 1 < s_hi) load_frame(obs + (size_t)(idx ? idx[s + 1] : s + 1) * FR);   // lands while this frame is multiplied
    const int g0 = (wave + 4 - ((s - s_lo) & 3)) & 3;          // this wave's groups: g0, g0 + 4, ... (the wave with seven of them rotates)
    const float* g = dy + ((size_t)s * 400 + 8 * h) * 32 + li;  // position 8h of group 0, channel li
    // dY rows of this wave's groups: TWO groups in flight (a group's eight values are split into their bf16 terms first, then the same registers take
    // the group after next) — with one group ahead the loop ran at the latency of these loads (122 us per 3840 frames)
    float bA[8], bB[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bA[j] = g[(g0 * 16 + j) * 32];
#pragma unroll
    for (int j = 0; j < 8; ++j) bB[j] = g[((g0 + 4) * 16 + j) * 32];          // g0 + 4 <= 7 < 25
    __syncthreads();
    const int glast = g0 + 4 * ((24 - g0) >> 2);      // this wave's last group
    auto group = [&](float (&bc)[8], int grp) __attribute__((always_inline)) {
      const int p0 = grp * 16 + 8 * h, oh0 = p0 / 20, ow0 = p0 - oh0 * 20;
      const unsigned short* fl = FB + kb + oh0 * 336 + ow0 * 4;
      const int wrap = 20 - ow0;                      // positions j >= wrap sit in the next output row
      c1_u32x4 b1, b2, b3, a[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t x1, x2, x3, y1, y2, y3;
        bs += bc[2 * j] + bc[2 * j + 1];
        if (C1W_ABL & 2) { x1 = x2 = x3 = __float_as_uint(bc[2 * j]); y1 = y2 = y3 = __float_as_uint(bc[2 * j + 1]); }
        else {
        c1_split3(bc[2 * j], x1, x2, x3);
        c1_split3(bc[2 * j + 1], y1, y2, y3);
        }
        b1[j] = __builtin_amdgcn_perm(y1, x1, 0x07060302u);
        b2[j] = __builtin_amdgcn_perm(y2, x2, 0x07060302u);
        b3[j] = __builtin_amdgcn_perm(y3, x3, 0x07060302u);
      }
      {   // unconditional (the group index is clamped): hipcc then knows how many loads are in flight and waits for exactly the older eight
        const int gn = (C1W_ABL & 1) ? grp : min(grp + 8, glast);
#pragma unroll
        for (int j = 0; j < 8; ++j) bc[j] = g[(gn * 16 + j) * 32];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int off0 = 8 * j + (2 * j >= wrap ? 336 - 80 : 0), off1 = 8 * j + 4 + (2 * j + 1 >= wrap ? 336 - 80 : 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int ko = (t >> 1) * 7056 + (t & 1) * 336;
          if (C1W_ABL & 4) a[t][j] = (uint32_t)(ko + off0 + grp) * 0x00010001u + (uint32_t)lane;
          else a[t][j] = (uint32_t)fl[ko + off0] | ((uint32_t)fl[ko + off1] << 16);
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (C1W_ABL & 8) { acc[t][0] += __uint_as_float(a[t][0] ^ b3[1]); continue; }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c1_bf16x8, a[t]), __builtin_bit_cast(c1_bf16x8, b3), acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (C1W_ABL & 8) { acc[t][0] += __uint_as_float(a[t][0] ^ b2[1]); continue; }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c1_bf16x8, a[t]), __builtin_bit_cast(c1_bf16x8, b2), acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (C1W_ABL & 8) { acc[t][0] += __uint_as_float(a[t][0] ^ b1[1]); continue; }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c1_bf16x8, a[t]), __builtin_bit_cast(c1_bf16x8, b1), acc[t], 0, 0, 0);
      }
    };
    int grp = g0;
#pragma unroll 1
    for (; grp + 4 < 25; grp += 8) {
      group(bA, grp);
      group(bB, grp + 4);
    }
    if (grp < 25) group(bA, grp);
  }
  // the four waves' sums: (1 -> 0, 3 -> 2), then 2 -> 0; a wave's 8 x 16 values per lane go through LDS as 32 pieces of 16 bytes, lane-contiguous
  float* R = reinterpret_cast<float*>(FB);
#pragma unroll 1
  for (int round = 0; round < ((C1W_ABL & 16) ? 0 : 2); ++round) {
    __syncthreads();
    const bool writer = round == 0 ? (wave & 1) : wave == 2;
    const bool reader = round == 0 ? !(wave & 1) : wave == 0;
    float* slot = R + (round == 0 ? (wave >> 1) : 0) * 8192 + lane * 4;
    if (writer) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(slot + (t * 4 + q) * 256) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    }
    __syncthreads();
    if (reader) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(slot + (t * 4 + q) * 256);
          acc[t][4 * q] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
        }
    }
  }
  // bias partial: every wave summed its own groups
  bs += __shfl_xor(bs, 32, 64);
  __syncthreads();
  if (h == 0) R[wave * 32 + li] = bs;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        part[((size_t)blockIdx.x * 256 + 32 * t + row) * 32 + li] = acc[t][e];
      }
    if (h == 0) bpart[blockIdx.x * 32 + li] = (R[li] + R[32 + li]) + (R[64 + li] + R[96 + li]);
  }
}

// The shipped exact weight gradient: the same work split on a DE-INTERLEAVED LDS frame (timing builds showed the eight 16-bit gathers per A fragment — 64 LDS instructions per group of 16
// positions — to cost 36 of the kernel's 127 us).  A tap (kh, kw) of output position (oh, ow) is pixel x[c][4 oh + kh][4 (ow + (kw >> 2)) + (kw & 3)]: with
// the frame stored by BYTE LANE, FBd[c][row][b = column & 3][d = column >> 2] (rows padded from 21 to 24), the positions ow .. ow+3 of one output row are
// four CONSECUTIVE bf16 for every tap, 8-byte aligned when ow is a multiple of 4; taps with kw >= 4 start one element later (a 32-bit funnel shift of
// the 8 + 4 bytes read, v_alignbit with a per-lane shift).  A group of 16 positions = two halves of (4 columns) x (2 output rows): 10 row pairs x 5
// column blocks = the frame's 25 groups exactly.  Per A fragment: 2 x (ds_read_b64 + ds_read_b32), bank-conflict-free, + 4 v_alignbit, instead of
// 8 x ds_read_u16 + 4 packs.  The conversion pass loads the frame as row-aligned quads of dwords (5 per 84-byte row + the 21st dword) so that a
// quad's byte lane b is four consecutive d: one ds_write_b64 per lane.
typedef uint32_t c1_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
#ifndef C1W_RS    // elements per (row, byte lane) of the de-interleaved frame: 21 used, a multiple of 4 (8-byte aligned fragments); 24 = 64,512 B, 28 = 75,264 B
#define C1W_RS 24
#endif
static constexpr int RS = C1W_RS, RSR = 4 * C1W_RS;   // RSR: elements per image row (four byte lanes)
__global__ __launch_bounds__(256, 2) void conv1_wgrad_exact_kernel(const uint8_t* obs, const int32_t* idx, const float* dy, float* part,
                                                                      float* bpart, int S, int frames_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned short FB[(4 * 84 * RSR > C1WX_LDS / 2 ? 4 * 84 * RSR : C1WX_LDS / 2)];   // FBd[4][84][4][RS] bf16; at least 64 KB for the closing reduction
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s_lo = blockIdx.x * frames_per_block, s_hi = min(S, s_lo + frames_per_block);
  if (s_lo >= s_hi) return;
  const int kb = ((li >> 3) * 4 + (li & 3)) * RS;   // tap (kh' = li >> 3, kw = li & 7): row kh', byte lane kw & 3; k-tile t adds (t >> 1) * 84 * RSR + (t & 1) * 4 * RSR
  const uint32_t sh = (li & 4) ? 16u : 0u;          // kw >= 4: one element later
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
  float bs = 0.0f;
  // conversion items: quads (c*84 + row, k = 0..4) -> dwords d = 4k .. 4k+3 of the row; singles (c*84 + row) -> dword d = 20
  c1_u32x4 pq[7];
  uint32_t p1[2];
  auto load_frame = [&](const uint8_t* frame) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int it = min(tid + 256 * i, 1679), cr = it / 5, k = it - cr * 5;
      pq[i] = *reinterpret_cast<const c1_u32x4_a4*>(frame + cr * 84 + 16 * k);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) p1[i] = *reinterpret_cast<const uint32_t*>(frame + min(tid + 256 * i, 335) * 84 + 80);
  };
  load_frame(obs + (size_t)(idx ? idx[s_lo] : s_lo) * FR);
  for (int s = s_lo; s < s_hi; ++s) {
    __syncthreads();  // previous frame fully consumed
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int it = tid + 256 * i, cr = it / 5, k = it - cr * 5;
      if (it < 1680) {
        unsigned short* dst = FB + cr * RSR + 4 * k;
        const uint32_t d0 = pq[i][0], d1 = pq[i][1], d2 = pq[i][2], d3 = pq[i][3];
        c1_u32x2 v;
        v[0] = c1_pack_hi16((float)(d0 & 255u), (float)(d1 & 255u)); v[1] = c1_pack_hi16((float)(d2 & 255u), (float)(d3 & 255u));
        *reinterpret_cast<c1_u32x2*>(dst) = v;
        v[0] = c1_pack_hi16((float)((d0 >> 8) & 255u), (float)((d1 >> 8) & 255u)); v[1] = c1_pack_hi16((float)((d2 >> 8) & 255u), (float)((d3 >> 8) & 255u));
        *reinterpret_cast<c1_u32x2*>(dst + RS) = v;
        v[0] = c1_pack_hi16((float)((d0 >> 16) & 255u), (float)((d1 >> 16) & 255u)); v[1] = c1_pack_hi16((float)((d2 >> 16) & 255u), (float)((d3 >> 16) & 255u));
        *reinterpret_cast<c1_u32x2*>(dst + 2 * RS) = v;
        v[0] = c1_pack_hi16((float)(d0 >> 24), (float)(d1 >> 24)); v[1] = c1_pack_hi16((float)(d2 >> 24), (float)(d3 >> 24));
        *reinterpret_cast<c1_u32x2*>(dst + 3 * RS) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cr = tid + 256 * i;
      if (cr < 336) {
        unsigned short* dst = FB + cr * RSR + 20;
        const uint32_t d = p1[i];
        dst[0] = (unsigned short)(__float_as_uint((float)(d & 255u)) >> 16);
        dst[RS] = (unsigned short)(__float_as_uint((float)((d >> 8) & 255u)) >> 16);
        dst[2 * RS] = (unsigned short)(__float_as_uint((float)((d >> 16) & 255u)) >> 16);
        dst[3 * RS] = (unsigned short)(__float_as_uint((float)(d >> 24)) >> 16);
      }
    }
    if (s + 1 < s_hi) load_frame(obs + (size_t)(idx ? idx[s + 1] : s + 1) * FR);   // lands while this frame is multiplied
    const int g0 = (wave + 4 - ((s - s_lo) & 3)) & 3;          // this wave's groups: g0, g0 + 4, ... (the wave with seven of them rotates)
    const int glast = g0 + 4 * ((24 - g0) >> 2);
    const float* g = dy + (size_t)s * 400 * 32 + li;
    // half hh = 2 grp + h: row pair hh / 5, column block hh % 5 -> positions (2 rp, 4 blk + j) for j < 4, (2 rp + 1, 4 blk + j - 4) for j >= 4
    auto half_pos = [&](int grp, int& rp, int& blk) __attribute__((always_inline)) { const int hh = 2 * grp + h; rp = hh / 5; blk = hh - rp * 5; };
    auto load_dy = [&](float (&bc)[8], int grp) __attribute__((always_inline)) {
      int rp, blk;
      half_pos(grp, rp, blk);
      const float* q = g + (rp * 40 + 4 * blk) * 32;
#pragma unroll
      for (int j = 0; j < 4; ++j) { bc[j] = q[j * 32]; bc[4 + j] = q[(20 + j) * 32]; }
    };
    float bA[8], bB[8];
    load_dy(bA, g0);
    load_dy(bB, g0 + 4);          // g0 + 4 <= 7 < 25
    __syncthreads();
    auto group = [&](float (&bc)[8], int grp) __attribute__((always_inline)) {
      int rp, blk;
      half_pos(grp, rp, blk);
      const unsigned short* fl = FB + kb + rp * (8 * RSR) + 4 * blk;
      c1_u32x4 b1, b2, b3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t x1, x2, x3, y1, y2, y3;
        bs += bc[2 * j] + bc[2 * j + 1];
        c1_split3(bc[2 * j], x1, x2, x3);
        c1_split3(bc[2 * j + 1], y1, y2, y3);
        b1[j] = __builtin_amdgcn_perm(y1, x1, 0x07060302u);
        b2[j] = __builtin_amdgcn_perm(y2, x2, 0x07060302u);
        b3[j] = __builtin_amdgcn_perm(y3, x3, 0x07060302u);
      }
      load_dy(bc, min(grp + 8, glast));       // unconditional (clamped): hipcc then knows how many loads are in flight
      // two k-tiles at a time (the kernel sits at the 256-register limit), the next pair's fragments requested before this pair's six MFMAs.
      // (sched_group_barrier hints that interleave single MFMAs with the next pair's LDS reads / shifts — and the same for the forward's conversions —
      // were measured: no gain, 110.3 vs 108.6 us and 90.2 vs 89.9; MFMA, VALU and LDS time add up on this chip whatever the order.)
      auto gather = [&](int th, c1_u32x4 (&a)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const unsigned short* ft = fl + ((th + t) >> 1) * (84 * RSR) + ((th + t) & 1) * (4 * RSR);
#pragma unroll
          for (int r = 0; r < 2; ++r) {         // the half's two output rows: 4 image rows = 4 * RSR elements apart
            const c1_u32x2 lo2 = *reinterpret_cast<const c1_u32x2*>(ft + r * 4 * RSR);
            const uint32_t hi1 = *reinterpret_cast<const uint32_t*>(ft + r * 4 * RSR + 4);
            a[t][2 * r] = __builtin_amdgcn_alignbit(lo2[1], lo2[0], sh);
            a[t][2 * r + 1] = __builtin_amdgcn_alignbit(hi1, lo2[1], sh);
          }
        }
      };
      c1_u32x4 af[2][2];
      gather(0, af[0]);
#pragma unroll
      for (int th = 0; th < 8; th += 2) {
        const int cur = (th >> 1) & 1, nxt = cur ^ 1;
        if (th + 2 < 8) gather(th + 2, af[nxt]);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[th + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c1_bf16x8, af[cur][t]), __builtin_bit_cast(c1_bf16x8, b3), acc[th + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[th + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c1_bf16x8, af[cur][t]), __builtin_bit_cast(c1_bf16x8, b2), acc[th + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[th + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c1_bf16x8, af[cur][t]), __builtin_bit_cast(c1_bf16x8, b1), acc[th + t], 0, 0, 0);
      }
    };
    int grp = g0;
#pragma unroll 1
    for (; grp + 4 < 25; grp += 8) {
      group(bA, grp);
      group(bB, grp + 4);
    }
    if (grp < 25) group(bA, grp);
  }
  // the four waves' sums: (1 -> 0, 3 -> 2), then 2 -> 0 (as in conv1_wgrad_exact_rowmajor_kernel)
  float* R = reinterpret_cast<float*>(FB);
#pragma unroll 1
  for (int round = 0; round < 2; ++round) {
    __syncthreads();
    const bool writer = round == 0 ? (wave & 1) : wave == 2;
    const bool reader = round == 0 ? !(wave & 1) : wave == 0;
    float* slot = R + (round == 0 ? (wave >> 1) : 0) * 8192 + lane * 4;
    if (writer) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(slot + (t * 4 + q) * 256) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    }
    __syncthreads();
    if (reader) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(slot + (t * 4 + q) * 256);
          acc[t][4 * q] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
        }
    }
  }
  bs += __shfl_xor(bs, 32, 64);
  __syncthreads();
  if (h == 0) R[wave * 32 + li] = bs;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        part[((size_t)blockIdx.x * 256 + 32 * t + row) * 32 + li] = acc[t][e];
      }
    if (h == 0) bpart[blockIdx.x * 32 + li] = (R[li] + R[32 + li]) + (R[64 + li] + R[96 + li]);
  }
}

// 3840-frame minibatch: 1280 blocks of 3 frames.  Alone 768 blocks of 5 frames were the fastest (225 us against 233 for 1024, and 25 MB of partials
// instead of 42), but beside the rollout the blocks of a grid that is resident all at once end 180-267 us after the first start (the CUs that also host
// actor blocks run theirs slower; tools/block_trace.py) and the kernel waits for the slowest: with a second, dynamically dispatched wave of shorter
// blocks the step is 0.3 ms shorter (pipelined 33.55 -> 33.23 ms, threaded 33.11 -> 32.82, three A/B rounds; 960 / 1024 / 1920 blocks: no gain).
#ifndef CBM_C1W_BLOCKS
#define CBM_C1W_BLOCKS 1280
#endif
static const int C1W_BLOCKS = CBM_C1W_BLOCKS;
int conv1_wgrad_frames_splits(int S) {
  // below 2048 frames (IMPALA's 21 x 30, the 1280-frame minibatches of a three-learner split) the round-4 rule stays: at most 768 blocks — those
  // launches fit one wave either way, and more partials only lengthen the fp32 chain of the partial reduction
  int blocks = S >= 2048 ? C1W_BLOCKS : (C1W_BLOCKS < 768 ? C1W_BLOCKS : 768);
  if (S < blocks) blocks = S;
  const int fpb = (S + blocks - 1) / blocks;
  return (S + fpb - 1) / fpb;
}
int conv1_wgrad_frames_splits_bound(int maxS) { return maxS < C1W_BLOCKS ? maxS : C1W_BLOCKS; }
void launch_conv1_wgrad_frames(const uint8_t* obs, const int32_t* idx, const float* dy, float* part, float* bpart, int S, hipStream_t st, bool split,
                               bool exact) {
  const int nz = conv1_wgrad_frames_splits(S);
  const int fpb = (S + nz - 1) / nz;
  static const bool dl = [] { const char* e = getenv("CBM_C1W_DL"); return !(e && e[0] == '0'); }();   // (=0: the row-major LDS frame with 16-bit gathers, A/B timing)
  if (exact && dl) hipLaunchKernelGGL(conv1_wgrad_exact_kernel, dim3(nz), dim3(256), 0, st, obs, idx, dy, part, bpart, S, fpb);
  else if (exact) hipLaunchKernelGGL(conv1_wgrad_exact_rowmajor_kernel, dim3(nz), dim3(256), 0, st, obs, idx, dy, part, bpart, S, fpb);
  else if (split) hipLaunchKernelGGL(conv1_wgrad_frames_split_kernel, dim3(nz), dim3(256), 0, st, obs, idx, dy, part, bpart, S, fpb);
  else hipLaunchKernelGGL(conv1_wgrad_frames_kernel, dim3(nz), dim3(256), 0, st, obs, idx, dy, part, bpart, S, fpb);
}
