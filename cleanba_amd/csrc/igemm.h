// igemm.h — one LDS-tiled implicit-GEMM kernel on v_mfma_f32_32x32x2_f32 for every
// GEMM-shaped op of the hot path (conv fwd / dgrad / wgrad, dense fwd / dgrad / wgrad).
//
//   C[x, y] = sum_r  Aop[x, r] * Bop[r, y]          x in [0,X), y in [0,Y), r in [r_lo, r_hi)
//
// The problem functor P supplies the gathers (im2col addressing, uint8->f32 conversion,
// transposed/flipped weights) and the epilogue; the kernel supplies tiling, double-buffered
// LDS staging with register prefetch, and the MFMA loop.
//
// Numerics: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain (MI355X guide §3),
// the K loop walks r in ascending order from a zero accumulator, and out-of-range r are
// staged as 0.0f (fma(0,b,acc) == acc), so C is exactly the chain the CPU oracle computes.
//
// Tiling: 256 threads = 4 waves arranged WX x WY; each wave owns TM x TN tiles of 32x32.
//   A tile in LDS: XR layout  As[x][r] pitch BR+1 (odd -> conflict-free ds_read_b32 when lanes
//                              walk x)            — global vectors run along r (fwd, dgrad)
//                  RX layout  As[r][x] pitch BX   — global vectors run along x (wgrad)
//   B tile in LDS: Bs[r][y] pitch BY; global vectors along y (RY) or along r (YR: transposed
//                  weight gather for dgrad).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef IG_BT_START   // timing builds (-DCBM_BLOCK_TRACE, gemm_layers.hip) stamp every block's start / end in igemm_dma_kernel
#define IG_BT_START() do { } while (0)
#define IG_BT_END() do { } while (0)
#endif
// Functors with KSKIP = true give every x-tile its own compressed reduction range: block_ctx(x0, cls) packs what the block needs,
// block_k(ctx) is the length of its range, r_map(ctx, r) turns a chunk-aligned compressed r into the real one (the chunk never
// straddles a 64-wide tap).  Used by the position-major dgrads, whose border tiles multiply only the taps that can be non-zero.
template <class P, class = void>
struct igemm_kskip { static constexpr bool value = false; };
template <class P>
struct igemm_kskip<P, decltype((void)P::KSKIP)> { static constexpr bool value = P::KSKIP; };
// Functors with BITMASK = true take their ReLU mask as bits: mask_word(x, y32, cls) is the 32-bit word of output row x for the 32 columns
// starting at y32, store_on(x, y, v, on, z, cls) the store.  Lane r of a wave fetches the word of tile row r ONCE and the epilogue hands
// it round with ds_bpermute, instead of every lane loading the fp32 activation behind each of its 16-32 outputs.
template <class P, class = void>
struct igemm_bitmask { static constexpr bool value = false; };
template <class P>
struct igemm_bitmask<P, decltype((void)P::BITMASK)> { static constexpr bool value = P::BITMASK; };
// Functors with MASKOUT = true (forward convs) emit their ReLU mask: store_flag() stores one output and returns "> 0", the epilogue
// ballots it, lane r of the lower wave half collects the word of tile row r, and put_mask() writes 32 words per 32x32 tile at once.
template <class P, class = void>
struct igemm_maskout { static constexpr bool value = false; };
template <class P>
struct igemm_maskout<P, decltype((void)P::MASKOUT)> { static constexpr bool value = P::MASKOUT; };
// Functors with ROWPTR = true split their gather addresses into a per-thread ROW part, computed once before the K loop, and a per-chunk
// part that is the same for the whole wave (scalar): element (x, r0 + rl) of A lives at a_origin() + a_off(x, rl, cls) + a_chunk(r0),
// element (r0 + rl, y) of B at b_origin() + b_off(rl, y, cls) + b_chunk(r0) (float offsets, 32-bit), valid while a chunk stays inside one
// contiguous run of the reduction index (one kernel row / one tap).  The generic load_a / load_b recompute frame / pixel / tap
// decompositions — integer divisions, clamps, 64-bit address math — for every 16-byte load of every chunk: 4-8 VALU instructions per
// MFMA in the conv kernels (rocprofv3 SQ_INSTS_VALU); with the split the K loop adds one scalar offset per load.
// Optional P::ROWEPI (with BITMASK): the rows of a 32-row output tile are a constant stride apart in the destination, so the functor hands the
// epilogue ONE pointer per (tile, lane) — EpiRow epi_row(x_tile0, y, cls) = {address of (tile row 0, column y), rows of the tile that exist (0 for a
// lane whose column does not), row stride in floats} — instead of decoding (frame, pixel) from x for each of the 16 values a lane stores
// (24-33 VALU instructions per stored value in the position-major dgrads, 15 % of the conv2 dgrad's time).
struct EpiRow { float* ptr; int valid; uint32_t stride; };
// 16-byte global load returned BY VALUE: `arr[j] = *reinterpret_cast<const float4*>(ptr)` is a struct copy that clang lowers to a
// memcpy(private <- global), which pins the whole register array in scratch memory (no SROA) — what made the row-pointer paths 50 % slower
static __device__ __forceinline__ float4 ig_ld4(const float* g) { const float4 v = *reinterpret_cast<const float4*>(g); return make_float4(v.x, v.y, v.z, v.w); }
template <class P, class = void>
struct igemm_rowepi { static constexpr bool value = false; };
template <class P>
struct igemm_rowepi<P, decltype((void)P::ROWEPI)> { static constexpr bool value = P::ROWEPI; };
// P::BIAS_PRE: the epilogue's bias is a function of the output column only — bias_pre(n) fetches it, store_pre(m, n, v, b) stores with it.  The
// small-batch kernel requests it in its prologue: fetched inside store() at the end, its L2 round trip was the last ~1 us of every block's 7 us life
// (tools/s16_trace.py).
template <class P, class = void>
struct igemm_bias_pre { static constexpr bool value = false; };
template <class P>
struct igemm_bias_pre<P, decltype((void)P::BIAS_PRE)> { static constexpr bool value = P::BIAS_PRE; };
template <class P, class = void>
struct igemm_rowptr_s16 { static constexpr bool value = false; };
template <class P>
struct igemm_rowptr_s16<P, decltype((void)P::ROWPTR_S16)> { static constexpr bool value = P::ROWPTR_S16; };
template <class P, class = void>
struct igemm_rowptr { static constexpr bool value = false; };
template <class P>
struct igemm_rowptr<P, decltype((void)P::ROWPTR)> { static constexpr bool value = P::ROWPTR; };

template <int BX_, int BY_, int BR_, int WX_, int WY_, int MINW_ = 4>
struct IgemmTile {
  static constexpr int BX = BX_, BY = BY_, BR = BR_, WX = WX_, WY = WY_, MINW = MINW_;  // MINW: min waves per SIMD (register budget)
  static_assert(WX_ * WY_ == 4, "4 waves per block");
  static_assert(BX_ % (32 * WX_) == 0 && BY_ % (32 * WY_) == 0 && BR_ % 4 == 0, "tile shape");
};

// P must provide:
//   using Tile = IgemmTile<...>;  static constexpr bool A_RX, B_YR, BIAS_GRAD;  static constexpr int NCLS;
//   int X() const, Y() const;  void r_range(int z, int& lo, int& hi) const;
//   float4 load_a(int x, int r, int rhi, int cls) const;   XR: A[x][r..r+3]   RX: A[x..x+3][r]
//   float4 load_b(int r, int y, int rhi, int cls) const;   RY: B[r][y..y+3]   YR: B[r..r+3][y]
//   void store(int x, int y, float v, int z, int cls) const;
//   void store_bias(int y, float v, int z) const;           (only if BIAS_GRAD)
template <class P>
__global__ __launch_bounds__(256, P::Tile::MINW) void igemm_kernel(const P p) {
  using T = typename P::Tile;
  constexpr int BX = T::BX, BY = T::BY, BR = T::BR, WX = T::WX, WY = T::WY;
  constexpr bool A_RX = P::A_RX, B_YR = P::B_YR;
  constexpr int TM = BX / WX / 32, TN = BY / WY / 32;
  constexpr int PA = A_RX ? BX : (BR + 1);
  constexpr int ASZ = A_RX ? BR * BX : BX * (BR + 1);
  constexpr int PB = BY;
  constexpr int BSZ = BR * PB;
  constexpr int NVA = (BX * BR / 4 + 255) / 256;
  constexpr int NVB = (BR * BY / 4 + 255) / 256;
  constexpr int RED = P::BIAS_GRAD ? 256 : 0;
  __shared__ __attribute__((aligned(16))) float smem[2 * ASZ + 2 * BSZ + RED];
  float* As = smem;
  float* Bs = smem + 2 * ASZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wx = wave / WY, wy = wave % WY;
  const int cls = P::NCLS > 1 ? (int)(blockIdx.y % P::NCLS) : 0;
  // XCD-aware tile order: block b runs on XCD b % 8 (each XCD has its own L2), and neighbouring x-tiles of an
  // im2col share input rows; give every XCD one contiguous run of x-tiles (bijective for any grid size).
  int bx = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, k = bx >> 3;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int x0 = bx * BX, y0 = (P::NCLS > 1 ? (int)(blockIdx.y / P::NCLS) : (int)blockIdx.y) * BY, z = blockIdx.z;
  int rlo, rhi;
  p.r_range(z, rlo, rhi);
  constexpr bool KSKIP = igemm_kskip<P>::value;
  int kctx = 0;
  if constexpr (KSKIP) { kctx = p.block_ctx(x0, cls); rlo = 0; rhi = p.block_k(kctx); }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  float4 ra[NVA], rb[NVB];
  float bsum = 0.0f;

  auto gload = [&](int rc) {
    int r0 = rc;
    if constexpr (KSKIP) r0 = p.r_map(kctx, rc);
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) {
        if (A_RX) {
          const int xq = v % (BX / 4), rl = v / (BX / 4);
          ra[j] = p.load_a(x0 + 4 * xq, r0 + rl, rhi, cls);
        } else {
          const int rq = v % (BR / 4), xl = v / (BR / 4);
          ra[j] = p.load_a(x0 + xl, r0 + 4 * rq, rhi, cls);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) {
        if (B_YR) {
          const int yl = v % BY, rq = v / BY;   // consecutive lanes -> consecutive y: the transposing LDS stores below are conflict-free
          rb[j] = p.load_b(r0 + 4 * rq, y0 + yl, rhi, cls);
        } else {
          const int yq = v % (BY / 4), rl = v / (BY / 4);
          rb[j] = p.load_b(r0 + rl, y0 + 4 * yq, rhi, cls);
        }
      }
    }
  };
  auto sstore = [&](int buf) {
    float* A_ = As + buf * ASZ;
    float* B_ = Bs + buf * BSZ;
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) {
        if (A_RX) {
          const int xq = v % (BX / 4), rl = v / (BX / 4);
          *reinterpret_cast<float4*>(A_ + rl * PA + 4 * xq) = ra[j];
        } else {
          const int rq = v % (BR / 4), xl = v / (BR / 4);
          float* d = A_ + xl * PA + 4 * rq;
          d[0] = ra[j].x; d[1] = ra[j].y; d[2] = ra[j].z; d[3] = ra[j].w;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) {
        if (B_YR) {
          const int yl = v % BY, rq = v / BY;
          float* d = B_ + (4 * rq) * PB + yl;
          d[0] = rb[j].x; d[PB] = rb[j].y; d[2 * PB] = rb[j].z; d[3 * PB] = rb[j].w;
        } else {
          const int yq = v % (BY / 4), rl = v / (BY / 4);
          *reinterpret_cast<float4*>(B_ + rl * PB + 4 * yq) = rb[j];
        }
      }
    }
  };

  gload(rlo);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int r0 = rlo; r0 < rhi; r0 += BR) {
    const bool more = (r0 + BR) < rhi;
    if (more) gload(r0 + BR);
    const float* A_ = As + buf * ASZ;
    const float* B_ = Bs + buf * BSZ;
    {
      // Fragment reads run one group (G k-pairs) ahead of the MFMAs.  hipcc otherwise sinks every ds_read next to
      // its MFMA (read, lgkmcnt(0), mfma, read, ...), which exposes one LDS round trip per 64-cycle MFMA; the
      // sched_barriers pin "all reads of group g+1, then the MFMAs of group g".
      constexpr int G = (BR / 2) % 4 == 0 ? 4 : 2, NG = BR / 2 / G;
      float fa[2][G][TM], fb[2][G][TN];
      auto frag = [&](int g, int set) {
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int rr = 2 * (g * G + q);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int xl = wx * (BX / WX) + i * 32 + li;
            fa[set][q][i] = A_RX ? A_[(rr + h) * PA + xl] : A_[xl * PA + rr + h];
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[set][q][j] = B_[(rr + h) * PB + wy * (BY / WY) + j * 32 + li];
        }
      };
      frag(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) frag(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][q][i], fb[g & 1][q][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (P::BIAS_GRAD) if (x0 == 0) {
      // column sums of the staged dY tile -> bias gradient partial (fixed order per thread)
      constexpr int PARTS = 256 / BY;
      const int yy = tid % BY, part = tid / BY;
      if (part < PARTS)
        for (int r = part; r < BR; r += PARTS) bsum += B_[r * PB + yy];
    }
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int y = y0 + wy * (BY / WY) + j * 32 + li;
      if constexpr (igemm_bitmask<P>::value) {
        const uint32_t mw = p.mask_word(x0 + wx * (BX / WX) + i * 32 + li, y0 + wy * (BY / WY) + j * 32, cls);   // lane r holds row r's word
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          const uint32_t w = (uint32_t)__shfl((int)mw, row, 64);
          p.store_on(x0 + wx * (BX / WX) + i * 32 + row, y, acc[i][j][e], (w >> li) & 1u, z, cls);
        }
      } else if constexpr (igemm_maskout<P>::value) {
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r0 = (e & 3) + 8 * (e >> 2);
          const bool on = p.store_flag(x0 + wx * (BX / WX) + i * 32 + r0 + 4 * h, y, acc[i][j][e], z, cls);
          const unsigned long long bal = __ballot(on);
          // lane r0 <- the lower wave half's row, lane r0 + 4 <- the upper half's (no writelane builtin in this clang)
          asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)bal), "n"(r0));
          asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)(bal >> 32)), "n"(r0 + 4));
        }
        if (lane < 32) p.put_mask(x0 + wx * (BX / WX) + i * 32 + lane, y0 + wy * (BY / WY) + j * 32, word);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          const int x = x0 + wx * (BX / WX) + i * 32 + row;
          p.store(x, y, acc[i][j][e], z, cls);
        }
      }
    }

  if constexpr (P::BIAS_GRAD) if (x0 == 0) {
    constexpr int PARTS = 256 / BY;
    float* red = smem + 2 * ASZ + 2 * BSZ;
    red[tid] = bsum;
    __syncthreads();
    if (tid < BY) {
      float s = red[tid];
      for (int q = 1; q < PARTS; ++q) s += red[q * BY + tid];
      p.store_bias(y0 + tid, s, z);
    }
  }
}

template <class P>
static inline void igemm_launch(const P& p, int nsplit, hipStream_t stream) {
  using T = typename P::Tile;
  dim3 grid((p.X() + T::BX - 1) / T::BX, ((p.Y() + T::BY - 1) / T::BY) * P::NCLS, nsplit);
  hipLaunchKernelGGL(igemm_kernel<P>, grid, dim3(256), 0, stream, p);
}

// ------------------------------------------------------------------------------------------------ two-chunk prefetch variant (dgrads)
// igemm_kernel keeps ONE K chunk in flight in registers while it multiplies the previous one.  With 16-wide chunks a wave's share of a
// chunk is only 8-16 MFMAs (0.2-0.4 us), less than a global-load round trip under load, so the staging store at the end of an iteration
// waits for its data.  Here two register sets alternate: the loads of chunk c+2 are issued before chunk c is multiplied and are not
// needed until the end of the NEXT iteration.  Same math, same k-ascending chain per accumulator -> bit-identical to igemm_kernel.
// Row-gather A, no fused bias gradient (the input-gradient problems); KSKIP and BITMASK functors supported.
template <class P>
__global__ __launch_bounds__(256, P::Tile::MINW) void igemm_pf2_kernel(const P p) {
  using T = typename P::Tile;
  constexpr int BX = T::BX, BY = T::BY, BR = T::BR, WX = T::WX, WY = T::WY;
  constexpr bool B_YR = P::B_YR;
  static_assert(!P::A_RX && !P::BIAS_GRAD, "two-chunk prefetch variant: dgrad-style problems");
  constexpr int TM = BX / WX / 32, TN = BY / WY / 32;
  constexpr int PA = BR + 1, ASZ = BX * PA, PB = BY, BSZ = BR * BY;
  constexpr int NVA = (BX * BR / 4 + 255) / 256, NVB = (BR * BY / 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * ASZ + 2 * BSZ];
  float* As = smem;
  float* Bs = smem + 2 * ASZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wx = wave / WY, wy = wave % WY;
  const int cls = P::NCLS > 1 ? (int)(blockIdx.y % P::NCLS) : 0;
  int bx = blockIdx.x, by = P::NCLS > 1 ? (int)(blockIdx.y / P::NCLS) : (int)blockIdx.y;
  if (P::NCLS == 1 && gridDim.y > 1 && (gridDim.x & 7) != 0) {
    // Many column tiles on a row count that is not a multiple of 8 (the dense input gradient: 30 x 49): workgroups go to the XCDs round robin by
    // LINEAR id x + NX * y, so "XCD = blockIdx.x % 8" (the remap below) only holds in the first row of the grid, and every XCD ended up fetching every
    // operand panel (206 MB of L2 misses for 14 MB of operands in round 6's counters).  Here the linear id is mapped onto the list ordered
    // (super-row of ceil(NX / 8) row tiles, column tile, row tile) and XCD j owns a contiguous run of it: its few A panels stay in its L2 for all columns.
    const int NX = gridDim.x, NY = gridDim.y, nb = NX * NY, L = (int)blockIdx.x + NX * (int)blockIdx.y;
    const int q = nb >> 3, r = nb & 7, xcd = L & 7, k = L >> 3;
    const int o = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    const int chunk = (NX + 7) >> 3, per = chunk * NY, sr = o / per, rem = o - sr * per, w = min(chunk, NX - sr * chunk);
    by = rem / w;
    bx = sr * chunk + (rem - by * w);
  } else {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, k = bx >> 3;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int x0 = bx * BX, y0 = by * BY, z = blockIdx.z;
  int rlo, rhi;
  p.r_range(z, rlo, rhi);
  constexpr bool KSKIP = igemm_kskip<P>::value;
  int kctx = 0;
  if constexpr (KSKIP) { kctx = p.block_ctx(x0, cls); rlo = 0; rhi = p.block_k(kctx); }
  const int nchunk = (rhi - rlo + BR - 1) / BR;


  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  constexpr bool RP = igemm_rowptr<P>::value;
  uint32_t arow[NVA], brow[NVB];
  if constexpr (RP) {
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j, rq = v % (BR / 4), xl = v / (BR / 4);
      arow[j] = p.a_off(x0 + xl, 4 * rq, cls);
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (B_YR) { const int yl = v % BY, rq = v / BY; brow[j] = p.b_off(4 * rq, y0 + yl, cls); }
      else { const int yq = v % (BY / 4), rl = v / (BY / 4); brow[j] = p.b_off(rl, y0 + 4 * yq, cls); }
    }
  }
  auto gload = [&](int c, float4 (&ra)[NVA], float4 (&rb)[NVB]) {
    int r0 = rlo + c * BR;
    if constexpr (KSKIP) r0 = p.r_map(kctx, r0);
    if constexpr (RP) {
      const uint32_t ao = p.a_chunk(r0), bo = p.b_chunk(r0);   // wave-uniform
#pragma unroll
      for (int j = 0; j < NVA; ++j)
        if (BX * BR / 4 % 256 == 0 || tid + 256 * j < BX * BR / 4) ra[j] = ig_ld4(p.a_origin() + (arow[j] + ao));
#pragma unroll
      for (int j = 0; j < NVB; ++j)
        if (BR * BY / 4 % 256 == 0 || tid + 256 * j < BR * BY / 4) rb[j] = ig_ld4(p.b_origin() + (brow[j] + bo));
    } else {   // (an early `return` out of the branch above keeps the register sets in scratch memory)
#pragma unroll
      for (int j = 0; j < NVA; ++j) {
        const int v = tid + 256 * j;
        if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) {
          const int rq = v % (BR / 4), xl = v / (BR / 4);
          ra[j] = p.load_a(x0 + xl, r0 + 4 * rq, rhi, cls);
        }
      }
#pragma unroll
      for (int j = 0; j < NVB; ++j) {
        const int v = tid + 256 * j;
        if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) {
          if (B_YR) { const int yl = v % BY, rq = v / BY; rb[j] = p.load_b(r0 + 4 * rq, y0 + yl, rhi, cls); }
          else { const int yq = v % (BY / 4), rl = v / (BY / 4); rb[j] = p.load_b(r0 + rl, y0 + 4 * yq, rhi, cls); }
        }
      }
    }
  };
  auto sstore = [&](int buf, const float4 (&ra)[NVA], const float4 (&rb)[NVB]) {
    float* A_ = As + buf * ASZ;
    float* B_ = Bs + buf * BSZ;
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) {
        const int rq = v % (BR / 4), xl = v / (BR / 4);
        float* d = A_ + xl * PA + 4 * rq;
        d[0] = ra[j].x; d[1] = ra[j].y; d[2] = ra[j].z; d[3] = ra[j].w;
      }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) {
        if (B_YR) { const int yl = v % BY, rq = v / BY; float* d = B_ + (4 * rq) * PB + yl; d[0] = rb[j].x; d[PB] = rb[j].y; d[2 * PB] = rb[j].z; d[3 * PB] = rb[j].w; }
        else { const int yq = v % (BY / 4), rl = v / (BY / 4); *reinterpret_cast<float4*>(B_ + rl * PB + 4 * yq) = rb[j]; }
      }
    }
  };
  auto compute = [&](int buf) {
    const float* A_ = As + buf * ASZ;
    const float* B_ = Bs + buf * BSZ;
    constexpr int G = (BR / 2) % 4 == 0 ? 4 : 2, NG = BR / 2 / G;
    float fa[2][G][TM], fb[2][G][TN];
    auto frag = [&](int g, int set) {
#pragma unroll
      for (int q = 0; q < G; ++q) {
        const int rr = 2 * (g * G + q);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[set][q][i] = A_[(wx * (BX / WX) + i * 32 + li) * PA + rr + h];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[set][q][j] = B_[(rr + h) * PB + wy * (BY / WY) + j * 32 + li];
      }
    };
    frag(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) frag(g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][q][i], fb[g & 1][q][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // (Round 6 measured the prefetches made unconditional — with `if (c + 2 < nchunk)` hipcc's wait in front of the staging store is the conservative
  // merge of two paths, `vmcnt(1)` / `vmcnt(0)` where `vmcnt(NV..)` would do (tools/isa_audit.py) — and it changed nothing: the loads of chunk c+2
  // are a whole chunk of MFMAs old by then.  dense / conv3 / conv2 input gradients 125.0 / 144.7 / 224.3 -> 126.8 / 145.4 / 223.5 us, the actor's
  // small-batch kernels slower (rollout alone 6.52 -> 6.7-6.8 ms: two more chunk loads per block).  Not kept: profiles/r06_isa_fixes_ab.txt.)
#ifndef PF2_ABL   // timing builds only (tools/variants.sh): 1 no epilogue stores, 2 no staging stores, 4 no global loads in the loop, 8 no MFMAs, 16 no barrier
#define PF2_ABL 0
#endif
  float4 a0[NVA], b0[NVB], a1[NVA], b1[NVB];
  gload(0, a0, b0);
  sstore(0, a0, b0);
  if (nchunk > 1) gload(1, a0, b0);
  if (PF2_ABL & 4) gload(0, a1, b1);
  __syncthreads();
  int buf = 0, c = 0;
  while (true) {
    if (!(PF2_ABL & 4) && c + 2 < nchunk) gload(c + 2, a1, b1);        // set 0 holds chunk c+1
    if (!(PF2_ABL & 8)) compute(buf);
    if (!(PF2_ABL & 2) && c + 1 < nchunk) sstore(buf ^ 1, a0, b0);
    if (!(PF2_ABL & 16)) __syncthreads();
    buf ^= 1;
    if (++c >= nchunk) break;
    if (!(PF2_ABL & 4) && c + 2 < nchunk) gload(c + 2, a0, b0);        // set 1 holds chunk c+1
    if (!(PF2_ABL & 8)) compute(buf);
    if (!(PF2_ABL & 2) && c + 1 < nchunk) sstore(buf ^ 1, a1, b1);
    if (!(PF2_ABL & 16)) __syncthreads();
    buf ^= 1;
    if (++c >= nchunk) break;
  }
  if (PF2_ABL & 1) {   // (the accumulators stay live through a sum that is stored on a condition no run meets)
    float sabl = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sabl += acc[i][j][e];
    if (sabl == 1.2345e-33f) p.store(x0, y0, sabl, z, cls);
    return;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int y = y0 + wy * (BY / WY) + j * 32 + li;
      if constexpr (igemm_bitmask<P>::value && igemm_rowepi<P>::value) {
        const uint32_t mw = p.mask_word(x0 + wx * (BX / WX) + i * 32 + li, y0 + wy * (BY / WY) + j * 32, cls);
        const EpiRow er = p.epi_row(x0 + wx * (BX / WX) + i * 32, y, cls);
        float* ph = er.ptr + (size_t)(4 * h) * er.stride;
        const int lim = er.valid - 4 * h;          // row r0 of this half-wave exists iff r0 < lim
        const uint32_t bit = 1u << li;
        if (__all(lim >= 28)) {                    // whole tile inside the problem (wave-uniform branch): no per-value bound test
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int r0 = (e & 3) + 8 * (e >> 2);
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)mw, r0), w1 = (uint32_t)__builtin_amdgcn_readlane((int)mw, r0 + 4);
            ph[(size_t)r0 * er.stride] = ((h ? w1 : w0) & bit) ? acc[i][j][e] : 0.0f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int r0 = (e & 3) + 8 * (e >> 2);
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)mw, r0), w1 = (uint32_t)__builtin_amdgcn_readlane((int)mw, r0 + 4);
            if (r0 < lim) ph[(size_t)r0 * er.stride] = ((h ? w1 : w0) & bit) ? acc[i][j][e] : 0.0f;
          }
        }
      } else if constexpr (igemm_bitmask<P>::value) {
        const uint32_t mw = p.mask_word(x0 + wx * (BX / WX) + i * 32 + li, y0 + wy * (BY / WY) + j * 32, cls);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          const uint32_t w = (uint32_t)__shfl((int)mw, row, 64);
          p.store_on(x0 + wx * (BX / WX) + i * 32 + row, y, acc[i][j][e], (w >> li) & 1u, z, cls);
        }
      } else if constexpr (igemm_maskout<P>::value) {
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r0 = (e & 3) + 8 * (e >> 2);
          const bool on = p.store_flag(x0 + wx * (BX / WX) + i * 32 + r0 + 4 * h, y, acc[i][j][e], z, cls);
          const unsigned long long bal = __ballot(on);
          asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)bal), "n"(r0));
          asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(word) : "s"((uint32_t)(bal >> 32)), "n"(r0 + 4));
        }
        if (lane < 32) p.put_mask(x0 + wx * (BX / WX) + i * 32 + lane, y0 + wy * (BY / WY) + j * 32, word);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          p.store(x0 + wx * (BX / WX) + i * 32 + row, y, acc[i][j][e], z, cls);
        }
      }
    }
}

template <class P>
static inline void igemm_pf2_launch(const P& p, int nsplit, hipStream_t stream) {
  using T = typename P::Tile;
  dim3 grid((p.X() + T::BX - 1) / T::BX, ((p.Y() + T::BY - 1) / T::BY) * P::NCLS, nsplit);
  hipLaunchKernelGGL(igemm_pf2_kernel<P>, grid, dim3(256), 0, stream, p);
}

// ------------------------------------------------------------------------------------------------ small-batch variant (actor steps)
// An actor step runs the network on 120 frames: 0.4-0.8 GFLOP per layer, a few microseconds of the chip.  On 64x64 tiles of
// v_mfma_f32_32x32x2_f32 that work lands on 90-375 blocks whose waves each own ONE accumulator and walk K in 16-wide chunks: 256-288
// MFMAs of 64 cycles back to back (6.8-7.7 us of one SIMD's matrix pipe) plus a barrier and an LDS round trip per chunk — 18-24 us per
// layer, eight launches per env-step, and the step time of the whole trainer grows by ~0.43 us for every microsecond of rollout.
// Here the same problems run on v_mfma_f32_16x16x4_f32 (k = 4 per instruction, 32 cycles): a wave owns a 16x16 output tile, so the same
// flops spread over 4x the waves (all 1024 SIMDs get work) and a wave's pipe time is K/4 x 32 cycles = 1.7-1.9 us; K chunks are 32-64
// wide (8-9 barriers per block instead of 32-36) with two chunks of loads in flight.  The instruction multiplies k = 4s .. 4s+3 in
// ascending order into the accumulator, steps ascend, chunks ascend: the SAME k-ascending fmaf chain as igemm_kernel -> identical bits
// (tests/test_gpu_parity.py::test_forward_bit_exact runs both).  Row-gather A / row-major B problems (the forward GEMMs), plain
// p.store epilogue (no ReLU-mask emission: actor workspaces have no masks).
//   LDS: A[x][r] pitch BR+4, B[r][y] unpadded with a column swizzle (see the kernel): both fragment reads (lane = (g4, r16): A[r16][4s+g4],
//   B[4s+g4][r16]) are conflict-free and both tiles are filled with 16-byte stores.
typedef float f32x4_mfma __attribute__((ext_vector_type(4)));
// timing build only (-DCBM_S16_TRACE; tools/s16_trace.py): clock stamps of the first, the middle and the last block of the launch whose X() equals
// cbm_s16_trace_sel (wave 0): entry, prologue, first loads issued / landed, every K chunk, end
#ifdef CBM_S16_TRACE
__device__ unsigned long long cbm_s16_trace[3][24];
__device__ int cbm_s16_trace_sel;
#define S16T(k) do { if (s16_tb >= 0 && (threadIdx.x & 255) == 0) cbm_s16_trace[s16_tb][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S16T(k) do { } while (0)
#endif
template <class P, int BX, int BY, int BR>
__global__ __launch_bounds__(256, 2) void igemm_s16_kernel(const P p) {
#ifdef CBM_S16_TRACE
  int s16_tb = -1;
  if (p.X() == cbm_s16_trace_sel && blockIdx.y == 0 && blockIdx.z == 0) {
    const int nb = gridDim.x;
    s16_tb = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == nb / 2 ? 1 : ((int)blockIdx.x == nb - 1 ? 2 : -1));
  }
#endif
  S16T(0);
  // (, 2): without it the register allocator aims at 8 waves per SIMD (64 VGPRs) and spills the prefetch sets to scratch — LDS caps these kernels at 3 blocks per CU anyway
  static_assert(BX % 16 == 0 && BY % 16 == 0 && BR % 32 == 0 && (BX / 16) * (BY / 16) % 4 == 0, "tile shape");
  static_assert(!P::A_RX && !P::B_YR && !P::BIAS_GRAD && P::NCLS == 1, "forward-style problems");
  constexpr int NT16 = (BX / 16) * (BY / 16) / 4;          // 16x16 tiles per wave
  // B rows are unpadded and the two halves of a row trade places on rows 1, 2 (mod 4) (XOR 16 on the column): of the four k rows of a fragment
  // read, g4 = 0 / 1 and g4 = 2 / 3 (the two 32-lane groups a ds_read_b32 is served in) sit on different halves of the 32 banks, like a pitch of
  // BY + 16 did — at 2/3 of the LDS (17.4 KB per block at BR = 32: a block now also fits beside the learner's conv2 weight-gradient blocks, which
  // leave 19 KB of a CU's LDS)
  static_assert(BY == 32, "the column swizzle of the B tile is written for 32-column tiles");
  constexpr int PA = BR + 4, PB = BY, ASZ = BX * PA, BSZ = BR * PB;
  constexpr int NVA = (BX * BR / 4 + 255) / 256, NVB = (BR * BY / 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * ASZ + 2 * BSZ];
  float* As = smem;
  float* Bs = smem + 2 * ASZ;
  // actor-size launches are latency-bound and share every SIMD with the learner's long-running waves: raised issue priority brings a concurrent
  // actor step from 3x to ~2.3x its isolated time (envpool-API path 327 k -> 349 k env-steps/s; the device-env step is unchanged, 36.15 -> 35.81 ms)
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  const int x0 = blockIdx.x * BX, y0 = blockIdx.y * BY, z = blockIdx.z;
  int rlo, rhi;
  p.r_range(z, rlo, rhi);
  const int nchunk = (rhi - rlo + BR - 1) / BR;
  // this wave's tiles: tile index q = wave * NT16 + i over the (BX/16) x (BY/16) grid, y fastest
  f32x4_mfma acc[NT16];
#pragma unroll
  for (int i = 0; i < NT16; ++i) acc[i] = f32x4_mfma{0.0f, 0.0f, 0.0f, 0.0f};

  // P::ROWPTR_S16: the row part of every gather address is computed once per thread, the chunk part is wave-uniform (SALU) — the functor's
  // load_a / load_b re-derive frame, pixel, tap and bounds for each 16-byte load (5-11 VALU instructions per MFMA in these short kernels)
  constexpr bool RP = igemm_rowptr_s16<P>::value;
  uint32_t arow[NVA], brow[NVB];
  if constexpr (RP) {
#pragma unroll
    for (int j = 0; j < NVA; ++j) { const int v = tid + 256 * j, rq = v % (BR / 4), xl = v / (BR / 4); arow[j] = p.a_off(x0 + xl, 4 * rq, 0); }
#pragma unroll
    for (int j = 0; j < NVB; ++j) { const int v = tid + 256 * j, yq = v % (BY / 4), rl = v / (BY / 4); brow[j] = p.b_off(rl, y0 + 4 * yq, 0); }
  }
  auto gload = [&](int c, float4 (&ra)[NVA], float4 (&rb)[NVB]) {
    const int r0 = rlo + c * BR;
    if constexpr (RP) {
      const uint32_t ao = p.a_chunk(r0), bo = p.b_chunk(r0);
#pragma unroll
      for (int j = 0; j < NVA; ++j)
        if (BX * BR / 4 % 256 == 0 || tid + 256 * j < BX * BR / 4) ra[j] = p.rp_a(arow[j] + ao);
#pragma unroll
      for (int j = 0; j < NVB; ++j)
        if (BR * BY / 4 % 256 == 0 || tid + 256 * j < BR * BY / 4) rb[j] = p.rp_b(brow[j] + bo);
    } else {   // (an early `return` out of the branch above keeps the register sets in scratch memory: no SROA)
#pragma unroll
      for (int j = 0; j < NVA; ++j) {
        const int v = tid + 256 * j;
        if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) { const int rq = v % (BR / 4), xl = v / (BR / 4); ra[j] = p.load_a(x0 + xl, r0 + 4 * rq, rhi, 0); }
      }
#pragma unroll
      for (int j = 0; j < NVB; ++j) {
        const int v = tid + 256 * j;
        if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) { const int yq = v % (BY / 4), rl = v / (BY / 4); rb[j] = p.load_b(r0 + rl, y0 + 4 * yq, rhi, 0); }
      }
    }
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
  if constexpr (igemm_bias_pre<P>::value) {
#pragma unroll
    for (int i = 0; i < NT16; ++i) { const int q = wave * NT16 + i, ty = q % (BY / 16); bpre[i] = p.bias_pre(y0 + ty * 16 + r16); }
  }
  S16T(1);
  gload(0, a0, b0);
  S16T(2);
  sstore(0, a0, b0);
  S16T(3);
  if (nchunk > 1) gload(1, a0, b0);
  __syncthreads();
  S16T(4);
  int buf = 0, c = 0;
  while (true) {
    if (c + 2 < nchunk) gload(c + 2, a1, b1);        // set 0 holds chunk c+1
    compute(buf);
    if (c + 1 < nchunk) sstore(buf ^ 1, a0, b0);
    __syncthreads();
    buf ^= 1;
    S16T(5 + (c < 16 ? c : 16));
    if (++c >= nchunk) break;
    if (c + 2 < nchunk) gload(c + 2, a0, b0);        // set 1 holds chunk c+1
    compute(buf);
    if (c + 1 < nchunk) sstore(buf ^ 1, a1, b1);
    __syncthreads();
    buf ^= 1;
    S16T(5 + (c < 16 ? c : 16));
    if (++c >= nchunk) break;
  }
  // D layout of the 16x16 tile: lane (g4, r16) holds rows 4*g4 + i, column r16
#pragma unroll
  for (int i = 0; i < NT16; ++i) {
    const int q = wave * NT16 + i, tx = q / (BY / 16), ty = q % (BY / 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (igemm_bias_pre<P>::value) p.store_pre(x0 + tx * 16 + 4 * g4 + e, y0 + ty * 16 + r16, acc[i][e], bpre[i]);
      else p.store(x0 + tx * 16 + 4 * g4 + e, y0 + ty * 16 + r16, acc[i][e], z, 0);
    }
  }
  S16T(22);
}
template <int BX, int BY, int BR, class P>
static inline void igemm_s16_launch(const P& p, int nsplit, hipStream_t stream) {
  dim3 grid((p.X() + BX - 1) / BX, (p.Y() + BY - 1) / BY, nsplit);
  hipLaunchKernelGGL((igemm_s16_kernel<P, BX, BY, BR>), grid, dim3(256), 0, stream, p);
}

// ------------------------------------------------------------------------------------------------ DMA-staged variant
// Same math and the same k-ascending accumulation order as igemm_kernel (bit-identical results), but the LDS tiles are filled by
// global_load_lds_dwordx4: the load unit writes 16 bytes per lane straight into LDS (wave-uniform base + lane*16) — no staging
// VGPRs, no ds_write instructions, no select/convert VALU work between the global load and the tile.  That fixes the tile layouts:
//   A[x][BR]       row-major, the BR/4 k-quads of row x stored in slot (quad ^ f(x)); the DMA writes lane-linear, so the swizzle is applied
//                  on the source side (lane l fetches the quad that belongs in slot l % QPR of row l / QPR).  One instruction touches
//                  64/QPR rows with QPR*16 contiguous bytes each — a quad-major A[r/4][x][4] tile (64 rows x 16 B per instruction)
//                  cost 4x the cache-line touches and measured 15 % slower on the dense layer.
//   B[r][y]        one instruction = 64 consecutive 16-byte pieces of the row-major [BR][BY] tile
// A fragments are ds_read_b32 (2-way bank conflict, inherent to reading one dword of each 16-byte granule; a ds_read_b128 +
// v_permlane32_swap variant was not faster — tools/ubench/gemm_dma.hip).  Usable when the gather needs no zero fill or conversion: P::DMA_OK, a_ptr(), b_ptr().
static __device__ __forceinline__ void ig_glds16(const float* g_lane, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g_lane, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                   0, 0);
}

template <class P>
__global__ __launch_bounds__(256, P::Tile::MINW) void igemm_dma_kernel(const P p) {
  using T = typename P::Tile;
  constexpr int BX = T::BX, BY = T::BY, BR = T::BR, WX = T::WX, WY = T::WY;
  static_assert(P::DMA_OK && !P::A_RX && !P::B_YR && !P::BIAS_GRAD && P::NCLS == 1 && BX % 64 == 0 && BY % 4 == 0 && (BR == 16 || BR == 32 || BR == 64), "DMA tile shape");
  constexpr int TM = BX / WX / 32, TN = BY / WY / 32;
  constexpr int ASZ = BR * BX, BSZ = BR * BY, PB = BY;
  constexpr int NIA = (BR / 4) * (BX / 64), NIB = BR * BY / 4 / 64;     // wave-instructions per tile
  static_assert(NIA % 4 == 0 && NIB % 4 == 0, "every wave issues the same number of DMA instructions");
  constexpr int NST = 3;                                                   // LDS stages: two chunks of copies in flight under the one being multiplied
  __shared__ __attribute__((aligned(16))) float smem[NST * (ASZ + BSZ)];
  float* As = smem;
  float* Bs = smem + NST * ASZ;
  IG_BT_START();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int wx = wave / WY, wy = wave % WY;
  // XCD-aware tile order.  Workgroups go to the eight XCDs round robin by LINEAR id, and each XCD has its own L2: the linear id (1-D grid over all
  // x and y tiles) is turned into "XCD j owns tiles [start_j, start_j + n_j) of the x-major list (g = x_tile * NY + y_tile)", so the NY column tiles of
  // an A row panel run on ONE XCD at the same time (the panel is fetched into one L2 once) and an XCD touches 1/8 of A.  (The round-4 form
  // remapped blockIdx.x alone: with 60 x-tiles, odd blockIdx.y rows land four XCDs further and every panel was fetched by two XCDs — 2.8x the
  // operand bytes in the PMC traffic.)
  const int NY = (p.Y() + BY - 1) / BY;
  int g = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = g & 7, k = g >> 3;
    g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int bx = g / NY, by = g - bx * NY;
  const int x0 = bx * BX, y0 = by * BY, z = blockIdx.z;
  int rlo, rhi;
  p.r_range(z, rlo, rhi);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  constexpr int QPR = BR / 4, RPI = 64 / QPR;            // k-quads per row, rows per wave-instruction
  auto swz = [](int row) { return (row / (16 / QPR)) & (QPR - 1); };
  auto dma = [&](int r0, int buf) {
    float* A_ = As + buf * ASZ;
    float* B_ = Bs + buf * BSZ;
#pragma unroll
    for (int i = 0; i < NIA / 4; ++i) {
      const int t = wave + 4 * i, row = t * RPI + lane / QPR, q = (lane % QPR) ^ swz(row);
      ig_glds16(p.a_ptr(x0 + row, r0 + 4 * q, 0), A_ + t * 256);
    }
#pragma unroll
    for (int i = 0; i < NIB / 4; ++i) {
      const int t = wave + 4 * i, u = t * 64 + lane, k = u / (BY / 4), n4 = u % (BY / 4);
      ig_glds16(p.b_ptr(r0 + k, y0 + 4 * n4, 0), B_ + t * 64 * 4);
    }
  };

  // Three-stage ring: chunk i is multiplied while the copies of chunks i+1 and i+2 are in flight (a chunk's 16 MFMAs per wave are ~0.45 us, an L2 /
  // fabric round trip under load is longer).  Each wave issues NIA/4 + NIB/4 copy instructions per chunk, in order, so "chunk i has landed" is
  // vmcnt(<= one chunk's instructions) while chunk i+1 is still outstanding; ONE barrier per chunk — it also says that every wave has left chunk
  // i-1, whose stage the copies of chunk i+2 overwrite.
  constexpr int NIW = NIA / 4 + NIB / 4;
  dma(rlo, 0);
  if (rlo + BR < rhi) dma(rlo + BR, 1);
  int buf = 0;
  for (int r0 = rlo; r0 < rhi; r0 += BR) {
    if (r0 + BR < rhi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                    // (not __syncthreads(): its fence would wait for every copy in flight)
#ifndef DMA_ABL   // timing builds only: 1 no epilogue stores, 2 no copies inside the K loop, 8 no MFMAs
#define DMA_ABL 0
#endif
    if (!(DMA_ABL & 2) && r0 + 2 * BR < rhi) dma(r0 + 2 * BR, buf >= 1 ? buf - 1 : NST - 1);      // stage (i + 2) % 3 = (buf + 2) % 3
    const float* A_ = As + buf * ASZ;
    const float* B_ = Bs + buf * BSZ;
    {
      constexpr int G = (BR / 2) % 4 == 0 ? 4 : 2, NG = BR / 2 / G;
      float fa[2][G][TM], fb[2][G][TN];
      auto frag = [&](int g, int set) {
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int rr = 2 * (g * G + q);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int row = wx * (BX / WX) + i * 32 + li;
            fa[set][q][i] = A_[row * BR + (((rr >> 2) ^ swz(row)) << 2) + (rr & 3) + h];
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[set][q][j] = B_[(rr + h) * PB + wy * (BY / WY) + j * 32 + li];
        }
      };
      frag(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) frag(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (DMA_ABL & 8) acc[i][j][0] += fa[g & 1][q][i] * fb[g & 1][q][j];
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][q][i], fb[g & 1][q][j], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
  }
  if (DMA_ABL & 1) {
    float sabl = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sabl += acc[i][j][e];
    if (sabl == 1.2345e-33f) p.store(x0, y0, sabl, z, 0);
    return;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int y = y0 + wy * (BY / WY) + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        p.store(x0 + wx * (BX / WX) + i * 32 + row, y, acc[i][j][e], z, 0);
      }
    }
  IG_BT_END();
}

template <class P>
static inline void igemm_dma_launch(const P& p, int nsplit, hipStream_t stream) {
  using T = typename P::Tile;
  dim3 grid(((p.X() + T::BX - 1) / T::BX) * ((p.Y() + T::BY - 1) / T::BY), 1, nsplit);   // 1-D over the tiles: the kernel orders them per XCD
  hipLaunchKernelGGL(igemm_dma_kernel<P>, grid, dim3(256), 0, stream, p);
}

// ------------------------------------------------------------------------------------------------ bf16 forward variant
// Build-only extension (the reference is fp32 everywhere): `cbm_config.forward_bf16` runs the forward GEMMs of conv2 /
// conv3 / dense on v_mfma_f32_32x32x16_bf16 — operands rounded to bf16 (RNE, v_cvt_pk_bf16_f32) when the tile is staged,
// fp32 accumulation, fp32 outputs.  Same functors, same global gathers; the LDS tiles hold bf16 with both operands
// k-contiguous (A[x][r], B[y][r], row pitch BR+8 halves = 80 B: 16-byte aligned and conflict-free for ds_read_b128), one
// 16-byte fragment per operand per MFMA.  Not bit-comparable with the fp32 chain: tolerance 2e-2 on logits.
typedef __bf16 ig_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ig_bf16x4 __attribute__((ext_vector_type(4)));

template <class P>
__global__ __launch_bounds__(256, P::Tile::MINW) void igemm_bf16_kernel(const P p) {
  using T = typename P::Tile;
  constexpr int BX = T::BX, BY = T::BY, BR = T::BR, WX = T::WX, WY = T::WY;
  static_assert(!P::A_RX && !P::BIAS_GRAD && P::NCLS == 1 && BR % 16 == 0, "bf16 variant covers the forward / dgrad style problems");
  constexpr bool B_YR = P::B_YR;
  constexpr int TM = BX / WX / 32, TN = BY / WY / 32;
  constexpr int PH = BR + 8;                       // halves per LDS row
  constexpr int ASZ = BX * PH, BSZ = BY * PH;      // halves per buffer
  constexpr int NVA = (BX * BR / 4 + 255) / 256;
  constexpr int NVB = (BR * BY / 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) __bf16 hmem[2 * ASZ + 2 * BSZ];
  __bf16* As = hmem;
  __bf16* Bs = hmem + 2 * ASZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wx = wave / WY, wy = wave % WY;
  int bx = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, k = bx >> 3;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int x0 = bx * BX, y0 = (int)blockIdx.y * BY, z = blockIdx.z;
  int rlo, rhi;
  p.r_range(z, rlo, rhi);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  float4 ra[NVA], rb[NVB];
  auto gload = [&](int r0) {
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) {
        const int rq = v % (BR / 4), xl = v / (BR / 4);
        ra[j] = p.load_a(x0 + xl, r0 + 4 * rq, rhi, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) {
        if (B_YR) {
          const int rq = v % (BR / 4), yl = v / (BR / 4);
          rb[j] = p.load_b(r0 + 4 * rq, y0 + yl, rhi, 0);
        } else {
          const int yq = v % (BY / 4), rl = v / (BY / 4);
          rb[j] = p.load_b(r0 + rl, y0 + 4 * yq, rhi, 0);
        }
      }
    }
  };
  auto cvt4 = [](float4 v) { ig_bf16x4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w; return o; };
  auto sstore = [&](int buf) {
    __bf16* A_ = As + buf * ASZ;
    __bf16* B_ = Bs + buf * BSZ;
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) {
        const int rq = v % (BR / 4), xl = v / (BR / 4);
        *reinterpret_cast<ig_bf16x4*>(A_ + xl * PH + 4 * rq) = cvt4(ra[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) {
        const ig_bf16x4 c = cvt4(rb[j]);
        if (B_YR) {
          const int rq = v % (BR / 4), yl = v / (BR / 4);
          *reinterpret_cast<ig_bf16x4*>(B_ + yl * PH + 4 * rq) = c;
        } else {  // global vector ran along y: transpose into B[y][r]
          const int yq = v % (BY / 4), rl = v / (BY / 4);
          __bf16* d = B_ + (4 * yq) * PH + rl;
          d[0] = c[0]; d[PH] = c[1]; d[2 * PH] = c[2]; d[3 * PH] = c[3];
        }
      }
    }
  };

  gload(rlo);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int r0 = rlo; r0 < rhi; r0 += BR) {
    const bool more = (r0 + BR) < rhi;
    if (more) gload(r0 + BR);
    const __bf16* A_ = As + buf * ASZ;
    const __bf16* B_ = Bs + buf * BSZ;
#pragma unroll
    for (int g = 0; g < BR / 16; ++g) {
      ig_bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const ig_bf16x8*>(A_ + (wx * (BX / WX) + i * 32 + li) * PH + 16 * g + 8 * h);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const ig_bf16x8*>(B_ + (wy * (BY / WY) + j * 32 + li) * PH + 16 * g + 8 * h);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int y = y0 + wy * (BY / WY) + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        p.store(x0 + wx * (BX / WX) + i * 32 + row, y, acc[i][j][e], z, 0);
      }
    }
}

template <class P>
static inline void igemm_bf16_launch(const P& p, int nsplit, hipStream_t stream) {
  using T = typename P::Tile;
  dim3 grid((p.X() + T::BX - 1) / T::BX, (p.Y() + T::BY - 1) / T::BY, nsplit);
  hipLaunchKernelGGL(igemm_bf16_kernel<P>, grid, dim3(256), 0, stream, p);
}

// ------------------------------------------------------------------------------------------------ split-bf16 variant (experiment, backward)
// fp32 products out of bf16 MFMAs: every fp32 operand is split exactly into NS bf16 terms when the tile is staged (v = v1 + v2 (+ v3) + r,
// |r| <= 2^-8NS |v|: the subtractions are exact), and the cross products that matter are formed on v_mfma_f32_32x32x16_bf16, 16x the
// fp32 MFMA rate: NS = 2 -> a1b1 + (a1b2 + a2b1), relative product error ~2^-16;  NS = 3 -> + (a2b2 + a1b3 + a3b1), ~2^-22, i.e. fp32
// rounding noise.  The leading product and the corrections accumulate in separate fp32 accumulators and are added once at the end.
// Same functors / gathers / epilogues as igemm_bf16_kernel (row-gather A, dgrad-style problems).
template <class P, int NS>
__global__ __launch_bounds__(256, 2) void igemm_split_kernel(const P p) {
  using T = typename P::Tile;
  constexpr int BX = T::BX, BY = T::BY, BR = T::BR, WX = T::WX, WY = T::WY;
  static_assert(!P::A_RX && !P::BIAS_GRAD && BR % 16 == 0 && (NS == 2 || NS == 3), "split variant covers the dgrad style problems");
  constexpr bool B_YR = P::B_YR;
  constexpr int TM = BX / WX / 32, TN = BY / WY / 32;
  constexpr int PH = BR + 8;
  constexpr int ASZ = BX * PH, BSZ = BY * PH, STG = NS * (ASZ + BSZ);
  constexpr int NVA = (BX * BR / 4 + 255) / 256;
  constexpr int NVB = (BR * BY / 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) __bf16 hmem[2 * STG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wx = wave / WY, wy = wave % WY;
  const int cls = P::NCLS > 1 ? (int)(blockIdx.y % P::NCLS) : 0;
  int bx = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, k = bx >> 3;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int x0 = bx * BX, y0 = (P::NCLS > 1 ? (int)(blockIdx.y / P::NCLS) : (int)blockIdx.y) * BY, z = blockIdx.z;
  int rlo, rhi;
  p.r_range(z, rlo, rhi);
  constexpr bool KSKIP = igemm_kskip<P>::value;
  int kctx = 0;
  if constexpr (KSKIP) { kctx = p.block_ctx(x0, cls); rlo = 0; rhi = p.block_k(kctx); }

  f32x16 acc[TM][TN], lo[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; lo[i][j][e] = 0.0f; }

  auto gload = [&](int rc, float4 (&ra)[NVA], float4 (&rb)[NVB]) {
    int r0 = rc;
    if constexpr (KSKIP) r0 = p.r_map(kctx, rc);
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) {
        const int rq = v % (BR / 4), xl = v / (BR / 4);
        ra[j] = p.load_a(x0 + xl, r0 + 4 * rq, rhi, cls);
      }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) {
        if (B_YR) { const int rq = v % (BR / 4), yl = v / (BR / 4); rb[j] = p.load_b(r0 + 4 * rq, y0 + yl, rhi, cls); }
        else { const int yq = v % (BY / 4), rl = v / (BY / 4); rb[j] = p.load_b(r0 + rl, y0 + 4 * yq, rhi, cls); }
      }
    }
  };
  // v -> NS bf16 planes (plane k of element e in out[k][e])
  auto split4 = [](float4 v, ig_bf16x4 (&out)[NS]) {
    float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r = f[e];
#pragma unroll
      for (int k = 0; k < NS; ++k) { const __bf16 t = (__bf16)r; out[k][e] = t; r -= (float)t; }
    }
  };
  auto sstore = [&](int buf, const float4 (&ra)[NVA], const float4 (&rb)[NVB]) {
    __bf16* base = hmem + buf * STG;
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (BX * BR / 4 % 256 == 0 || v < BX * BR / 4) {
        const int rq = v % (BR / 4), xl = v / (BR / 4);
        ig_bf16x4 pl[NS];
        split4(ra[j], pl);
#pragma unroll
        for (int k = 0; k < NS; ++k) *reinterpret_cast<ig_bf16x4*>(base + k * ASZ + xl * PH + 4 * rq) = pl[k];
      }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (BR * BY / 4 % 256 == 0 || v < BR * BY / 4) {
        ig_bf16x4 pl[NS];
        split4(rb[j], pl);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          __bf16* B_ = base + NS * ASZ + k * BSZ;
          if (B_YR) { const int rq = v % (BR / 4), yl = v / (BR / 4); *reinterpret_cast<ig_bf16x4*>(B_ + yl * PH + 4 * rq) = pl[k]; }
          else {
            const int yq = v % (BY / 4), rl = v / (BY / 4);
            __bf16* d = B_ + (4 * yq) * PH + rl;
            d[0] = pl[k][0]; d[PH] = pl[k][1]; d[2 * PH] = pl[k][2]; d[3 * PH] = pl[k][3];
          }
        }
      }
    }
  };

  auto compute = [&](int buf) {
    const __bf16* Ab = hmem + buf * STG;
    const __bf16* Bb = Ab + NS * ASZ;
#pragma unroll
    for (int g = 0; g < BR / 16; ++g) {
      ig_bf16x8 fa[NS][TM], fb[NS][TN];
#pragma unroll
      for (int k = 0; k < NS; ++k) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[k][i] = *reinterpret_cast<const ig_bf16x8*>(Ab + k * ASZ + (wx * (BX / WX) + i * 32 + li) * PH + 16 * g + 8 * h);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[k][j] = *reinterpret_cast<const ig_bf16x8*>(Bb + k * BSZ + (wy * (BY / WY) + j * 32 + li) * PH + 16 * g + 8 * h);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
          if constexpr (NS == 3) {
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], lo[i][j], 0, 0, 0);
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], lo[i][j], 0, 0, 0);
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], lo[i][j], 0, 0, 0);
          }
          lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], lo[i][j], 0, 0, 0);
          lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], lo[i][j], 0, 0, 0);
        }
    }
  };
  // two register sets alternate: the loads of chunk c+2 are issued before chunk c is multiplied (the MFMA work of a chunk is far shorter
  // than a global-load round trip here, so one chunk of lookahead left every iteration waiting for its data)
  const int nchunk = (rhi - rlo + BR - 1) / BR;
  float4 a0[NVA], b0[NVB], a1[NVA], b1[NVB];
  gload(rlo, a0, b0);
  sstore(0, a0, b0);
  if (nchunk > 1) gload(rlo + BR, a0, b0);
  __syncthreads();
  int buf = 0, c = 0;
  while (true) {
    if (c + 2 < nchunk) gload(rlo + (c + 2) * BR, a1, b1);
    compute(buf);
    if (c + 1 < nchunk) sstore(buf ^ 1, a0, b0);
    __syncthreads();
    buf ^= 1;
    if (++c >= nchunk) break;
    if (c + 2 < nchunk) gload(rlo + (c + 2) * BR, a0, b0);
    compute(buf);
    if (c + 1 < nchunk) sstore(buf ^ 1, a1, b1);
    __syncthreads();
    buf ^= 1;
    if (++c >= nchunk) break;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int y = y0 + wy * (BY / WY) + j * 32 + li;
      if constexpr (igemm_bitmask<P>::value) {
        const uint32_t mw = p.mask_word(x0 + wx * (BX / WX) + i * 32 + li, y0 + wy * (BY / WY) + j * 32, cls);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          const uint32_t w = (uint32_t)__shfl((int)mw, row, 64);
          p.store_on(x0 + wx * (BX / WX) + i * 32 + row, y, acc[i][j][e] + lo[i][j][e], (w >> li) & 1u, z, cls);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          p.store(x0 + wx * (BX / WX) + i * 32 + row, y, acc[i][j][e] + lo[i][j][e], z, cls);
        }
      }
    }
}

template <class P, int NS>
static inline void igemm_split_launch(const P& p, int nsplit, hipStream_t stream) {
  using T = typename P::Tile;
  dim3 grid((p.X() + T::BX - 1) / T::BX, ((p.Y() + T::BY - 1) / T::BY) * P::NCLS, nsplit);
  hipLaunchKernelGGL((igemm_split_kernel<P, NS>), grid, dim3(256), 0, stream, p);
}

// Weight-gradient flavour of the split-bf16 kernel: C[x][y] = sum_r A[r][x] * B[r][y] with both operands r-major in memory (x / y
// contiguous), i.e. the MFMA's k index runs ACROSS global rows.  Every thread takes 4x4 blocks (4 consecutive r, 4 consecutive x or y:
// four 16-byte loads), transposes them in registers and writes r-contiguous 8-byte pieces, so the tiles land as A[x][r] / B[y][r] without
// 2-byte scatter stores.  K chunk = 32 rows.  BIAS_GRAD (column sums of B for the x0 == 0 blocks) is taken from the fp32 registers
// before the split.  Functors: ConvWgrad / MatWgrad (A_RX, !B_YR).
template <class P, int NS>
__global__ __launch_bounds__(256, 2) void igemm_split_wgrad_kernel(const P p) {
  using T = typename P::Tile;
  constexpr int BX = T::BX, BY = T::BY, WX = T::WX, WY = T::WY, KR = 32;
  static_assert(P::A_RX && !P::B_YR && P::NCLS == 1 && (NS == 2 || NS == 3), "split wgrad variant");
  constexpr int TM = BX / WX / 32, TN = BY / WY / 32;
  constexpr int PH = KR + 8;
  constexpr int ASZ = BX * PH, BSZ = BY * PH, STG = NS * (ASZ + BSZ);
  constexpr int NBA = BX / 4 * (KR / 4), NBB = BY / 4 * (KR / 4);       // 4x4 blocks per tile
  constexpr int NVA = (NBA + 255) / 256, NVB = (NBB + 255) / 256;
  __shared__ __attribute__((aligned(16))) __bf16 hmem[2 * STG];
  __shared__ float bred[P::BIAS_GRAD ? 256 * 4 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wx = wave / WY, wy = wave % WY;
  int bx = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, k = bx >> 3;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int x0 = bx * BX, y0 = (int)blockIdx.y * BY, z = blockIdx.z;
  int rlo, rhi;
  p.r_range(z, rlo, rhi);

  f32x16 acc[TM][TN], lo[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; lo[i][j][e] = 0.0f; }

  float4 ra[NVA][4], rb[NVB][4];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto gload = [&](int r0) {
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (NBA % 256 == 0 || v < NBA) {
        const int rq = v % (KR / 4), xq = v / (KR / 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[j][i] = p.load_a(x0 + 4 * xq, r0 + 4 * rq + i, rhi, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (NBB % 256 == 0 || v < NBB) {
        const int rq = v % (KR / 4), yq = v / (KR / 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[j][i] = p.load_b(r0 + 4 * rq + i, y0 + 4 * yq, rhi, 0);
      }
    }
  };
  auto comp = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
  // 4x4 block (rows = 4 consecutive r, columns = 4 consecutive x|y) -> for each column one r-contiguous quad, split into NS planes
  auto put_block = [&](const float4 (&blk)[4], __bf16* plane0, int plane_stride, int col0, int rq) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float r[4] = {comp(blk[0], c), comp(blk[1], c), comp(blk[2], c), comp(blk[3], c)};
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        ig_bf16x4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const __bf16 t = (__bf16)r[e]; q[e] = t; r[e] -= (float)t; }
        *reinterpret_cast<ig_bf16x4*>(plane0 + k * plane_stride + (col0 + c) * PH + 4 * rq) = q;
      }
    }
  };
  auto sstore = [&](int buf) {
    __bf16* base = hmem + buf * STG;
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + 256 * j;
      if (NBA % 256 == 0 || v < NBA) put_block(ra[j], base, ASZ, 4 * (v / (KR / 4)), v % (KR / 4));
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + 256 * j;
      if (NBB % 256 == 0 || v < NBB) {
        put_block(rb[j], base + NS * ASZ, BSZ, 4 * (v / (KR / 4)), v % (KR / 4));
        if constexpr (P::BIAS_GRAD) if (x0 == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { bsum.x += rb[j][i].x; bsum.y += rb[j][i].y; bsum.z += rb[j][i].z; bsum.w += rb[j][i].w; }
        }
      }
    }
  };

  gload(rlo);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int r0 = rlo; r0 < rhi; r0 += KR) {
    const bool more = (r0 + KR) < rhi;
    if (more) gload(r0 + KR);
    const __bf16* Ab = hmem + buf * STG;
    const __bf16* Bb = Ab + NS * ASZ;
#pragma unroll
    for (int g = 0; g < KR / 16; ++g) {
      ig_bf16x8 fa[NS][TM], fb[NS][TN];
#pragma unroll
      for (int k = 0; k < NS; ++k) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[k][i] = *reinterpret_cast<const ig_bf16x8*>(Ab + k * ASZ + (wx * (BX / WX) + i * 32 + li) * PH + 16 * g + 8 * h);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[k][j] = *reinterpret_cast<const ig_bf16x8*>(Bb + k * BSZ + (wy * (BY / WY) + j * 32 + li) * PH + 16 * g + 8 * h);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
          if constexpr (NS == 3) {
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], lo[i][j], 0, 0, 0);
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], lo[i][j], 0, 0, 0);
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], lo[i][j], 0, 0, 0);
          }
          lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], lo[i][j], 0, 0, 0);
          lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], lo[i][j], 0, 0, 0);
        }
    }
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int y = y0 + wy * (BY / WY) + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        p.store(x0 + wx * (BX / WX) + i * 32 + row, y, acc[i][j][e] + lo[i][j][e], z, 0);
      }
    }
  if constexpr (P::BIAS_GRAD) if (x0 == 0) {
    // thread v < NBB holds the sums of columns 4*yq..4*yq+3 over its r-quads; add the KR/4 threads of a column group in rq order
    reinterpret_cast<float4*>(bred)[tid] = bsum;
    __syncthreads();
    if (tid < BY) {
      const int yq = tid >> 2, c = tid & 3;
      float s = 0.0f;
      for (int rq = 0; rq < KR / 4; ++rq) s += bred[(yq * (KR / 4) + rq) * 4 + c];
      p.store_bias(y0 + tid, s, z);
    }
  }
}

template <class P, int NS>
static inline void igemm_split_wgrad_launch(const P& p, int nsplit, hipStream_t stream) {
  using T = typename P::Tile;
  dim3 grid((p.X() + T::BX - 1) / T::BX, (p.Y() + T::BY - 1) / T::BY, nsplit);
  hipLaunchKernelGGL((igemm_split_wgrad_kernel<P, NS>), grid, dim3(256), 0, stream, p);
}
