// pointwise.hip — the HBM-stream / scan kernels of the hot path: action sampling, GAE, advantage
// normalisation, jax-compatible permutation, PPO and IMPALA(V-trace) loss heads, Adam / RMSProp.
// Reference lines are cited per kernel ("ppo" = cleanba_ppo.py, "impala" = cleanba_impala.py).
#include "cbm_internal.h"
#include "ppo_loss.h"
#include <math.h>
#include <float.h>

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------
// ppo:256-261 / impala:296-300.  32 lanes per env (one per action): u = uniform(subkey,[B,A]) via threefry
// counters, Gumbel-max argmax (first max wins) by a shuffle tournament, log_softmax at the chosen action with the
// exp-sum accumulated in ascending action order (same order as the oracle -> same bits).
__global__ __launch_bounds__(256) void sample_kernel(const float* logits, int B, int A, uint32_t sk0, uint32_t sk1, int32_t* actions,
                                                     float* logprobs, const float* value_in, float* value_out, float* logits_out) {
  const int b = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int a = threadIdx.x & 31;
  const bool live = b < B && a < A;
  const int bb = b < B ? b : B - 1, aa = a < A ? a : A - 1;
  const float z = logits[(size_t)bb * A + aa];
  const uint32_t n = (uint32_t)(B * A);
  const float u = cbm_bits_to_uniform(cbm_random_bits_at(sk0, sk1, n, (uint32_t)(bb * A + aa)));
  float g = live ? z - cbm_logf(-cbm_logf(u)) : -INFINITY;
  if (live && logits_out) logits_out[(size_t)b * A + a] = z;
  // argmax with first-max-wins: a strictly greater value wins, ties go to the lower index
  int bi = a;
  float bv = g, mx = live ? z : -INFINITY;
  for (int o = 16; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 32);
    const int oi = __shfl_xor(bi, o, 32);
    const float om = __shfl_xor(mx, o, 32);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    mx = om > mx ? om : mx;
  }
  // all 32 lanes of the group now agree on (bv, bi, mx)
  const float e = live ? cbm_expf(z - mx) : 0.0f;
  const float zb = __shfl(z, bi, 32);
  float s = 0.0f;
  for (int j = 0; j < A; ++j) s += __shfl(e, j, 32);
  if (b < B && a == 0) {
    actions[b] = bi;
    if (logprobs) logprobs[b] = (zb - mx) - cbm_logf(s);
    if (value_out) value_out[b] = value_in[b];
  }
}
void launch_sample(const float* logits, int B, int A, uint32_t sk0, uint32_t sk1, int32_t* actions, float* logprobs,
                   const float* value_in, float* value_out, float* logits_out, hipStream_t st) {
  hipLaunchKernelGGL(sample_kernel, dim3(ceil_div(B, 8)), dim3(256), 0, st, logits, B, A, sk0, sk1, actions, logprobs, value_in,
                     value_out, logits_out);
}

// ------------------------------------------------------------------------------------------
// compute_gae ppo:532-560: one thread per env column, serial reverse scan over T.
__global__ void gae_kernel(const float* rewards, const float* values, const uint8_t* dones, const float* next_value,
                           const uint8_t* next_done, int T, int B, float gamma, float gl, float* adv, float* target) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float a = 0.0f;
  float nv = next_value[b];
  float nd = (float)next_done[b];
  // sixteen steps' operands are requested before the first of them is used: with the loads inside the recursion every step waited for its own
  // round trip to L2 (50 us for T = 128); the recursion itself is unchanged
  for (int t1 = T; t1 > 0; t1 -= 16) {
    float v[16], r[16], d[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = max(t1 - 1 - u, 0);
      v[u] = values[(size_t)t * B + b]; r[u] = rewards[(size_t)t * B + b]; d[u] = (float)dones[(size_t)t * B + b];
    }
    const int t1v = cbm_opaque_vgpr(t1);   // (a per-lane copy: the guard below is then a predicate, not a scalar branch per step)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = t1v - 1 - u;
      if (t >= 0) {
        const float nnt = 1.0f - nd;
        const float delta = (r[u] + (gamma * nv) * nnt) - v[u];
        a = delta + ((gl * nnt) * a);
        adv[(size_t)t * B + b] = a;
        target[(size_t)t * B + b] = a + v[u];
        nv = v[u];
        nd = d[u];
      }
    }
  }
}
void launch_gae(const float* rewards, const float* values, const uint8_t* dones, const float* next_value, const uint8_t* next_done,
                int T, int B, float gamma, float lambda, float* adv, float* target, hipStream_t st) {
  const float gl = (float)((double)gamma * (double)lambda);
  hipLaunchKernelGGL(gae_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, rewards, values, dones, next_value, next_done, T, B, gamma, gl,
                     adv, target);
}

// ------------------------------------------------------------------------------------------
// ppo:592-595: per column-group mean / population std over (T, B/G); wavefront + LDS reduction.
__device__ float block_sum_1024(float v, float* red) {
  // fixed-order tree: wave shuffle (64 lanes) then 16 wave totals
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float s = 0.0f;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}
__global__ __launch_bounds__(1024) void advnorm_kernel(float* adv, int T, int B, int groups) {
  __shared__ float red[16];
  const int g = blockIdx.x, w = B / groups, n = T * w;
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += adv[(size_t)(i / w) * B + g * w + (i % w)];
  const float mean = block_sum_1024(s, red) / (float)n;
  float v = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { const float d = adv[(size_t)(i / w) * B + g * w + (i % w)] - mean; v += d * d; }
  const float sd = sqrtf(block_sum_1024(v, red) / (float)n);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const size_t o = (size_t)(i / w) * B + g * w + (i % w);
    adv[o] = (adv[o] - mean) / (sd + 1e-8f);
  }
}
void launch_advnorm(float* adv, int T, int B, int groups, hipStream_t st) {
  hipLaunchKernelGGL(advnorm_kernel, dim3(groups), dim3(1024), 0, st, adv, T, B, groups);
}

// ------------------------------------------------------------------------------------------
// rlax.vtrace_td_error_and_advantage (impala:559-567) on its own, lambda = 1, clip thresholds 1: one thread per env column, the serial
// reverse recursion and then the q / pg-advantage pass — the same expressions, in the same order, as the loop inside impala_loss_kernel
// and as the oracle, so the three agree bit for bit.  Used by the parity tests (cbm_vtrace); the training path keeps the fused kernel.
__global__ void vtrace_kernel(const float* v_tm1, const float* v_t, const float* r_t, const float* disc_t, const float* rho_tm1, int T, int B,
                              float* errors, float* pg_adv, float* q_est) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float e = 0.0f;
  for (int t = T - 1; t >= 0; --t) {
    const size_t i = (size_t)t * B + b;
    const float cr = rho_tm1[i] < 1.0f ? rho_tm1[i] : 1.0f;
    const float td = cr * ((r_t[i] + disc_t[i] * v_t[i]) - v_tm1[i]);
    e = td + (disc_t[i] * cr) * e;
    errors[i] = (e + v_tm1[i]) - v_tm1[i];
  }
  for (int t = 0; t < T; ++t) {
    const size_t i = (size_t)t * B + b;
    const float cr = rho_tm1[i] < 1.0f ? rho_tm1[i] : 1.0f;
    const float qb = (t == T - 1) ? v_t[i] : (errors[i + B] + v_tm1[i + B]);
    const float q = r_t[i] + disc_t[i] * qb;
    q_est[i] = q;
    pg_adv[i] = cr * (q - v_tm1[i]);
  }
}
void launch_vtrace(const float* v_tm1, const float* v_t, const float* r_t, const float* disc_t, const float* rho_tm1, int T, int B, float* errors,
                   float* pg_adv, float* q_est, hipStream_t st) {
  hipLaunchKernelGGL(vtrace_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, v_tm1, v_t, r_t, disc_t, rho_tm1, T, B, errors, pg_adv, q_est);
}

// ------------------------------------------------------------------------------------------
// Async rollouts (legacy `--async-batch-size`): prepare_data's reward re-index (naturecnn:232-255) + env-id-indexed compute_gae
// (naturecnn:467-531) in one pass.  One block per env: 1024 rows at a time (from the end), each thread looks its env up in one row's B
// ids and parks (value, done, reward, flat index) in LDS; thread 0 then runs the serial recursion over the parked samples.  The "next
// sample of the same env" the reference finds through next_index_ranges is simply the sample visited just before in this reverse walk:
// its reward / done / value are the carries.  An env's last sample gets delta = 0 and, with the done carry starting at 1, advantage 0.
#define GA_ROWS 1024
__global__ __launch_bounds__(GA_ROWS) void gae_async_kernel(const int32_t* env_ids, const float* rewards, const float* values, const uint8_t* dones,
                                                            int R, int B, float gamma, float gl, float* adv, float* target) {
  __shared__ float sv[GA_ROWS], sr[GA_ROWS];
  __shared__ int32_t si[GA_ROWS];   // flat index << 1 | done, or -1 when the env is not in that row
  const int e = blockIdx.x;
  float nv = 0.0f, nd = 1.0f, nr = 0.0f, a = 0.0f;
  bool first = true;
  for (int rhi = R; rhi > 0; rhi -= GA_ROWS) {
    const int r = rhi - 1 - (int)threadIdx.x;   // thread 0 holds the latest row of the chunk
    int32_t tag = -1;
    if (r >= 0) {
      const int32_t* row = env_ids + (size_t)r * B;
      int c = -1;
      for (int j = 0; j < B; ++j) c = row[j] == e ? j : c;
      if (c >= 0) {
        const int i = r * B + c;
        sv[threadIdx.x] = values[i];
        sr[threadIdx.x] = rewards[i];
        tag = (i << 1) | (dones[i] ? 1 : 0);
      }
    }
    si[threadIdx.x] = tag;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int n = rhi < GA_ROWS ? rhi : GA_ROWS;
      for (int k = 0; k < n; ++k) {
        const int32_t t = si[k];
        if (t < 0) continue;
        const int i = t >> 1;
        const float v = sv[k];
        const float nnt = 1.0f - nd;
        const float delta = first ? 0.0f : (nr + (gamma * nv) * nnt) - v;
        a = delta + ((gl * nnt) * a);
        adv[i] = a;
        target[i] = a + v;
        nv = v; nd = (float)(t & 1); nr = sr[k];
        first = false;
      }
    }
    __syncthreads();
  }
}
void launch_gae_async(const int32_t* env_ids, const float* rewards, const float* values, const uint8_t* dones, int R, int B, int num_envs,
                      float gamma, float lambda, float* adv, float* target, hipStream_t st) {
  const float gl = (float)((double)gamma * (double)lambda);
  hipLaunchKernelGGL(gae_async_kernel, dim3(num_envs), dim3(GA_ROWS), 0, st, env_ids, rewards, values, dones, R, B, gamma, gl, adv, target);
}

// naturecnn:540-541: the legacy ppo_loss normalises the advantages of each minibatch (population std).  out[idx[i]] gets the normalised
// value, so the loss kernel keeps reading "adv[n]" — minibatches partition the permutation, the scattered writes never collide.
__global__ __launch_bounds__(1024) void mb_advnorm_kernel(const float* adv, const int32_t* idx, int n, float* out) {
  __shared__ float red[16];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += adv[idx ? idx[i] : i];
  const float mean = block_sum_1024(s, red) / (float)n;
  float v = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { const float d = adv[idx ? idx[i] : i] - mean; v += d * d; }
  const float sd = sqrtf(block_sum_1024(v, red) / (float)n);
  for (int i = threadIdx.x; i < n; i += blockDim.x) { const int k = idx ? idx[i] : i; out[k] = (adv[k] - mean) / (sd + 1e-8f); }
}
void launch_mb_advnorm(const float* adv, const int32_t* idx, int n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(mb_advnorm_kernel, dim3(1), dim3(1024), 0, st, adv, idx, n, out);
}

// ------------------------------------------------------------------------------------------
// jax.random.permutation (ppo:606): per round, composite key (random_bits << 32 | position) makes the sort stable by construction; an element's
// place in the sorted order is the number of keys below it.  Counting against all n keys was n^2 = 2.4e8 64-bit compares (32 us per round, two rounds
// per epoch: 360 us of every update); the keys are uniform, so they are first dealt into PERM_G ranges by their top bits (a block ranks its 256
// elements per range in LDS and reserves its run in each range's list with ONE global atomic per range: list order is arbitrary, the COUNT below a
// key is not), and an element is then counted against its own range only — n / 64 keys — plus the sizes of the ranges below.  The same pass
// writes out[place] = in[position]: no rank array, no separate scatter.  Scratch (uint64 units): PERM_G lists of n keys, then the PERM_G counters.
#define PERM_G 64
#define PERM_LOG2G 6
size_t permutation_scratch_u64(int n) { return (size_t)PERM_G * (size_t)n + 64; }
__device__ __forceinline__ void perm_bucket_body(uint32_t sk0, uint32_t sk1, int n, uint64_t* lists, int32_t* counts) {
  __shared__ int32_t hist[PERM_G], base[PERM_G];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < PERM_G) hist[threadIdx.x] = 0;
  __syncthreads();
  uint64_t key = 0;
  int g = 0, local = 0;
  if (i < n) {
    const uint32_t bits = cbm_random_bits_at(sk0, sk1, (uint32_t)n, (uint32_t)i);
    key = ((uint64_t)bits << 32) | (uint32_t)i;
    g = (int)(bits >> (32 - PERM_LOG2G));
    local = atomicAdd(&hist[g], 1);
  }
  __syncthreads();
  if (threadIdx.x < PERM_G) base[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], hist[threadIdx.x]) : 0;
  __syncthreads();
  if (i < n) lists[(size_t)g * n + base[g] + local] = key;
}
// block (x, g): elements [256 x, +256) of range g's list; place = (sizes of the ranges below) + #{keys of the range below mine}
__device__ __forceinline__ void perm_place_body(const uint64_t* lists, const int32_t* counts, int n, const int32_t* in_vals, int32_t* out_vals) {
  __shared__ uint64_t tile[512];
  __shared__ int32_t cs[PERM_G];
  const int g = blockIdx.y, cnt = counts[g];
  if ((int)blockIdx.x * 256 >= cnt) return;
  if (threadIdx.x < PERM_G) cs[threadIdx.x] = threadIdx.x < g ? counts[threadIdx.x] : 0;
  const uint64_t* list = lists + (size_t)g * n;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const uint64_t mine = e < cnt ? list[e] : 0;
  int r = 0;
  for (int j0 = 0; j0 < cnt; j0 += 512) {
    __syncthreads();
    for (int q = threadIdx.x; q < 512; q += 256) tile[q] = (j0 + q < cnt) ? list[j0 + q] : ~0ull;
    __syncthreads();
    const int m = (min(512, cnt - j0) + 7) & ~7;      // (the pad entries compare as "not below")
    for (int q = 0; q < m; q += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) r += tile[q + u] < mine ? 1 : 0;
    }
  }
  int below = 0;
#pragma unroll
  for (int h = 0; h < PERM_G; ++h) below += cs[h];
  if (e < cnt) { const int i = (int)(uint32_t)mine; out_vals[below + r] = in_vals ? in_vals[i] : i; }
}
__global__ __launch_bounds__(256) void perm_bucket_kernel(uint32_t sk0, uint32_t sk1, int n, uint64_t* lists, int32_t* counts) {
  perm_bucket_body(sk0, sk1, n, lists, counts);
}
__global__ __launch_bounds__(256) void perm_place_kernel(const uint64_t* lists, const int32_t* counts, int n, const int32_t* in_vals, int32_t* out_vals) {
  perm_place_body(lists, counts, n, in_vals, out_vals);
}
static int perm_rounds(int n) {
  const double sz = n > 1 ? (double)n : 1.0;
  return (int)ceil(3.0 * log(sz) / log(4294967295.0));
}
void launch_permutation(const uint32_t key_in[2], int n, int32_t* perm, int32_t* tmp, uint64_t* scratch, hipStream_t st) {
  uint32_t k0 = key_in[0], k1 = key_in[1];
  const int rounds = perm_rounds(n);
  int32_t* counts = reinterpret_cast<int32_t*>(scratch + (size_t)PERM_G * n);
  int32_t* bufs[2] = {perm, tmp};
  int cur = (rounds % 2 == 1) ? 0 : 1;  // ping-pong so the final round lands in `perm`
  const int32_t* src = nullptr;
  for (int r = 0; r < rounds; ++r) {
    uint32_t n0, n1, s0, s1;
    cbm_split_at(k0, k1, 2, 0, &n0, &n1);
    cbm_split_at(k0, k1, 2, 1, &s0, &s1);
    k0 = n0; k1 = n1;
    hipMemsetAsync(counts, 0, PERM_G * sizeof(int32_t), st);
    hipLaunchKernelGGL(perm_bucket_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, s0, s1, n, scratch, counts);
    hipLaunchKernelGGL(perm_place_kernel, dim3(ceil_div(n, 256), PERM_G), dim3(256), 0, st, scratch, counts, n, src, bufs[cur]);
    src = bufs[cur];
    cur ^= 1;
  }
}
// The permutations of ALL epochs of an update (ppo:599-606: one subkey per epoch, known before the first minibatch) in four launches instead of
// six per epoch: job (epoch e, round r) has its own lists and counters, one bucket pass takes every job (blockIdx.y), one place pass per round
// takes that round of every epoch (blockIdx.z).  The same kernels' bodies on the same keys: the same permutations.  perms / tmps: [ne][n].
struct PermKeys { uint32_t k[CBM_PERM_BATCH_MAX][2]; };
__global__ __launch_bounds__(256) void perm_bucket_batch_kernel(const PermKeys keys, int n, uint64_t* lists, int32_t* counts) {
  const int job = blockIdx.y;
  perm_bucket_body(keys.k[job][0], keys.k[job][1], n, lists + (size_t)job * PERM_G * n, counts + job * PERM_G);
}
__global__ __launch_bounds__(256) void perm_place_batch_kernel(const uint64_t* lists, const int32_t* counts, int n, int rounds, int r,
                                                               const int32_t* in_base, int32_t* out_base) {
  const int e = blockIdx.z, job = e * rounds + r;
  perm_place_body(lists + (size_t)job * PERM_G * n, counts + job * PERM_G, n, in_base ? in_base + (size_t)e * n : nullptr, out_base + (size_t)e * n);
}
// does launch_permutations_batch take (n, ne)?  (the contexts that do size their buffers for it, the others keep the one-epoch sizes)
bool permutation_batch_ok(int n, int ne) {
  const int rounds = perm_rounds(n);
  return ne >= 2 && rounds >= 1 && ne * rounds <= CBM_PERM_BATCH_MAX;
}
size_t permutation_batch_scratch_u64(int n, int ne) {
  const size_t jobs = (size_t)ne * perm_rounds(n);
  return jobs * PERM_G * (size_t)n + jobs * PERM_G / 2 + 64;
}
bool launch_permutations_batch(const uint32_t (*epoch_keys)[2], int ne, int n, int32_t* perms, int32_t* tmps, uint64_t* scratch, hipStream_t st) {
  const int rounds = perm_rounds(n), jobs = ne * rounds;
  if (!permutation_batch_ok(n, ne)) return false;
  PermKeys keys;
  for (int e = 0; e < ne; ++e) {
    uint32_t k0 = epoch_keys[e][0], k1 = epoch_keys[e][1];
    for (int r = 0; r < rounds; ++r) {   // launch_permutation's key chain
      uint32_t n0, n1, s0, s1;
      cbm_split_at(k0, k1, 2, 0, &n0, &n1);
      cbm_split_at(k0, k1, 2, 1, &s0, &s1);
      k0 = n0; k1 = n1;
      keys.k[e * rounds + r][0] = s0; keys.k[e * rounds + r][1] = s1;
    }
  }
  int32_t* counts = reinterpret_cast<int32_t*>(scratch + (size_t)jobs * PERM_G * n);
  hipMemsetAsync(counts, 0, (size_t)jobs * PERM_G * sizeof(int32_t), st);
  hipLaunchKernelGGL(perm_bucket_batch_kernel, dim3(ceil_div(n, 256), jobs), dim3(256), 0, st, keys, n, scratch, counts);
  int32_t* bufs[2] = {perms, tmps};
  int cur = (rounds % 2 == 1) ? 0 : 1;
  const int32_t* src = nullptr;
  for (int r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(perm_place_batch_kernel, dim3(ceil_div(n, 256), PERM_G, ne), dim3(256), 0, st, scratch, counts, n, rounds, r, src, bufs[cur]);
    src = bufs[cur];
    cur ^= 1;
  }
  return true;
}

// ------------------------------------------------------------------------------------------
// PPO loss head ppo:516-577: per-sample statistics + analytic dL/dlogits, dL/dvalue into dzv[N][32].
// 32 lanes per sample, one per action: the exponentials run in parallel, the three softmax sums are taken in ascending action order by a
// shuffle walk (the same order, hence the same bits, as a serial loop and as the oracle), and every lane writes its own dzv column
// (one 128-byte row per sample).  Eight samples per block; the block adds its samples' statistics in sample order.
__global__ __launch_bounds__(256) void ppo_loss_kernel(const float* logits, const float* value, int N, int A, const int32_t* idx,
                                                        const int32_t* actions, const float* old_logprob, const float* adv,
                                                        const float* target, float clip_coef, float ent_coef, float vf_coef,
                                                        float* dzv, float* partials) {
  __shared__ float red[8][4];
  const int g = threadIdx.x >> 5, j = threadIdx.x & 31;
  const int i = blockIdx.x * 8 + g;
  const bool live = i < N;
  const int ii = live ? i : N - 1;
  const int n = idx ? idx[ii] : ii;
  const int a = actions[n];
  const float invN = 1.0f / (float)N;
  const float zj = logits[(size_t)ii * A + (j < A ? j : A - 1)];
  PpoSampleStats ss;
  const float d = ppo_loss_lane(zj, j, A, a, value[ii], old_logprob[n], adv[n], target[n], clip_coef, ent_coef, vf_coef, invN, ss);
  if (live) dzv[(size_t)i * 32 + j] = d;
  if (j == 0) {
    red[g][0] = live ? ss.pg : 0.0f; red[g][1] = live ? ss.dv2 : 0.0f; red[g][2] = live ? ss.ent : 0.0f;
    red[g][3] = live ? ss.kl : 0.0f;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.0f;
    for (int q = 0; q < 8; ++q) v += red[q][threadIdx.x];
    partials[blockIdx.x * 4 + threadIdx.x] = v;
  }
}
// sum of the block partials: 64 lanes take strided subsets in ascending order, then a fixed shuffle tree
__global__ void ppo_stats_kernel(const float* partials, int nblk, int N, float ent_coef, float vf_coef, float* stats5) {
  ppo_stats_wave(partials, nblk, N, ent_coef, vf_coef, stats5);
}
void launch_ppo_stats(const float* partials, int nblk, int N, float ent_coef, float vf_coef, float* stats5, hipStream_t st) {
  hipLaunchKernelGGL(ppo_stats_kernel, dim3(1), dim3(64), 0, st, partials, nblk, N, ent_coef, vf_coef, stats5);
}
void launch_ppo_loss(const float* logits, const float* value, int N, int A, const int32_t* idx, const int32_t* actions,
                     const float* old_logprob, const float* adv, const float* target, float clip_coef, float ent_coef, float vf_coef,
                     float* dzv, float* partials, float* stats5, hipStream_t st) {
  const int nblk = ceil_div(N, 8);
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(nblk), dim3(256), 0, st, logits, value, N, A, idx, actions, old_logprob, adv, target, clip_coef,
                     ent_coef, vf_coef, dzv, partials);
  hipLaunchKernelGGL(ppo_stats_kernel, dim3(1), dim3(64), 0, st, partials, nblk, N, ent_coef, vf_coef, stats5);
}

// ------------------------------------------------------------------------------------------
// IMPALA loss head impala:569-597 + rlax 0.1.5 V-trace (lambda = 1, rho/c/pg clips = 1).
// One BLOCK per env column of the minibatch (network outputs are [T1][Bm] rows, t-major; storage fields are [T1][ld] with this
// minibatch at columns col0..col0+Bm).  Only Bm = 30 blocks exist, so the kernel is as long as one block's instruction stream: the work is
// cut into phases that are either ELEMENT-parallel (one thread per (t, action): the 3 x T x A exponentials, the probabilities, the gradients)
// or STEP-parallel (one thread per t: maxima, the sums over actions in ascending action order, logarithms), every operand in LDS, and the
// V-trace recursion itself runs on one lane while the other waves take the entropy sums.  No loop bound or guard is a scalar branch: bounds
// that are uniform (A, T) are compared through a VGPR copy so that the compiler predicates instead — with `if (j < A)` inside unrolled loops
// the one-step-per-thread form of this kernel spent most of its 40-47 us in s_cbranch (tools/il_trace.py: scan 9.3 us for 128 steps).
// Every per-element expression and every summation order is the one of impala_loss_column_kernel below (one thread per column, the
// form the oracle restates), so results are bit-identical to it.
#ifdef CBM_IL_TRACE   // timing build: phase stamps of block 0, thread 0 (read back by tools/il_trace.py through cbm_debug_il_trace)
__device__ unsigned long long cbm_il_trace[16];
extern "C" int cbm_debug_il_trace(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(cbm_il_trace), sizeof(cbm_il_trace)) == hipSuccess ? 0 : -1; }
#define ILT(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) cbm_il_trace[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ILT(k) do { } while (0)
#endif
#define IL_NT 512
#define IL_PAD 32          // floats behind each [T][A] array / in front of the scan arrays: unguarded reads and writes land here
// sum of row[0..A) in ascending order; all 28 candidates are read before the first add (row has IL_PAD floats of slack behind the array)
static __device__ __forceinline__ float il_row_sum(const float* row, int Av) {
  float v[28];
#pragma unroll
  for (int q = 0; q < 28; ++q) v[q] = row[q];
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < 28; ++q) { const float n = s + v[q]; s = q < Av ? n : s; }
  return s;
}
static __device__ __forceinline__ float il_row_max(const float* row, int Av) {
  float v[28];
#pragma unroll
  for (int q = 0; q < 28; ++q) v[q] = row[q];
  float mx = v[0];
#pragma unroll
  for (int q = 1; q < 28; ++q) mx = (q < Av && v[q] > mx) ? v[q] : mx;
  return mx;
}
size_t impala_loss_lds_bytes(int T1, int A) { return ((size_t)24 * T1 + 3 * IL_PAD + 3 * ((size_t)(T1 - 1) * A + IL_PAD)) * sizeof(float); }
__global__ __launch_bounds__(IL_NT) void impala_loss_kernel(const float* logits, const float* value, const float* mu_logits, const int32_t* actions,
                                   const float* rewards, const uint8_t* dones, const uint8_t* firststeps, int T1, int Bm, int A, int col0,
                                   int ld, float gamma, float vf_coef, float ent_coef, float* dzv, float* partials) {
  extern __shared__ float ish[];
  const int b = blockIdx.x, T = T1 - 1, tid = threadIdx.x, TA = T * A;
  const int Av = cbm_opaque_vgpr(A), Tv = cbm_opaque_vgpr(T);
  float* p = ish;
  auto take = [&](int n) { float* r = p; p += n; return r; };
  float* s_val = take(T1);                 // value[t][b], t = 0..T
  int* s_act = (int*)take(T1);
  float* s_disc = take(T1);                // (1 - done) * gamma
  float* s_rew = take(T1);
  float* s_mask = take(T1);                // 1 - firststep
  float* s_mx = take(2 * T1);              // [0, T): max_j z   [T, 2T): max_j mu
  float* s_ma = take(T1);                  // mu logit of the action taken
  float* s_lpa = take(T1);                 // log pi(a)
  float* s_lma = take(T1);                 // log mu(a)
  float* s_se = take(T1);                  // sum_j exp(z_j - mx)
  float* s_lse = take(T1);                 // its logarithm
  float* s_cr = take(T1);                  // min(1, rho)
  float* s_H = take(T1);                   // entropy
  float* s_pg = take(T1);                  // per-t loss terms, summed in t order at the end
  float* s_bl = take(T1);
  float* s_en = take(T1);
  float* s_c1 = take(T1);                  // -pgadv * mask
  float* s_c2 = take(T1);                  // ent_coef * mask
  float* s_dv = take(T1);                  // dL/dvalue
  float* s_td = take(T1 + IL_PAD) + IL_PAD;    // cr * (r + disc*v' - v); the scan reads up to 15 entries below [0]
  float* s_dc = take(T1 + IL_PAD) + IL_PAD;    // disc * cr
  float* s_err = take(T1 + IL_PAD) + IL_PAD;   // err recursion
  float* zs = take(TA + IL_PAD);           // z[t][j]
  float* ez = take(TA + IL_PAD);           // exp(z - mx)
  float* ms = take(TA + IL_PAD);           // mu[t][j] -> exp(mu - mm) -> p_j * log p_j
  ILT(0);
  // element index e = t*A + j of this thread's k-th element: e = tid + k*IL_NT, walked without a division per element
  const int t_first = tid / A, j_first = tid - t_first * A, dt = IL_NT / A, dj = IL_NT - dt * A;
  // phase 0: operands into LDS
  {
    int t = t_first, j = j_first;
    for (int e = tid; e < TA; e += IL_NT) {
      zs[e] = logits[((size_t)t * Bm + b) * A + j];
      ms[e] = mu_logits[((size_t)t * ld + col0 + b) * A + j];
      t += dt; j += dj;
      const bool wrap = j >= Av; j = wrap ? j - A : j; t = wrap ? t + 1 : t;
    }
    for (int t1 = tid; t1 < T1; t1 += IL_NT) {
      s_val[t1] = value[(size_t)t1 * Bm + b];
      if (t1 < T) {
        const size_t sidx = (size_t)t1 * ld + col0 + b;
        s_act[t1] = actions[sidx];
        s_disc[t1] = (1.0f - (float)dones[sidx]) * gamma;
        s_rew[t1] = rewards[sidx];
        s_mask[t1] = 1.0f - (float)firststeps[sidx];
      }
    }
  }
  __syncthreads();
  ILT(1);
  // phase 1 (one thread per (policy, t)): row maxima; the behaviour policy's side also keeps mu[a] (its row is overwritten in phase 2)
  for (int u = tid; u < 2 * T; u += IL_NT) {
    const bool mu = u >= T;
    const int t = mu ? u - T : u;
    const float* row = (mu ? ms : zs) + t * A;
    s_mx[u] = il_row_max(row, Av);
    if (mu) s_ma[t] = row[s_act[t]];
  }
  __syncthreads();
  ILT(2);
  // phase 2 (per element): the exponentials of both softmaxes
  {
    int t = t_first, j = j_first;
    for (int e = tid; e < TA; e += IL_NT) {
      ez[e] = cbm_expf(zs[e] - s_mx[t]);
      ms[e] = cbm_expf(ms[e] - s_mx[T + t]);
      t += dt; j += dj;
      const bool wrap = j >= Av; j = wrap ? j - A : j; t = wrap ? t + 1 : t;
    }
  }
  __syncthreads();
  ILT(3);
  // phase 3 (one thread per (policy, t)): sums in ascending action order, logarithms, log pi(a) / log mu(a)
  for (int u = tid; u < 2 * T; u += IL_NT) {
    const bool mu = u >= T;
    const int t = mu ? u - T : u;
    const float sum = il_row_sum((mu ? ms : ez) + t * A, Av);
    const float lg = cbm_logf(sum);
    if (mu) s_lma[t] = (s_ma[t] - s_mx[u]) - lg;
    else { s_lpa[t] = (zs[t * A + s_act[t]] - s_mx[u]) - lg; s_se[t] = sum; s_lse[t] = lg; }
  }
  __syncthreads();
  ILT(4);
  // phase 4: the scan's inputs (one thread per t), and p_j * log p_j per element into the freed mu rows
  for (int t = tid; t < T; t += IL_NT) {
    const float rho = cbm_expf(s_lpa[t] - s_lma[t]);
    const float cr = rho < 1.0f ? rho : 1.0f;
    const float disc = s_disc[t];
    s_cr[t] = cr;
    s_td[t] = cr * ((s_rew[t] + disc * s_val[t + 1]) - s_val[t]);
    s_dc[t] = disc * cr;
  }
  {
    int t = t_first, j = j_first;
    for (int e = tid; e < TA; e += IL_NT) {
      const float lp = (zs[e] - s_mx[t]) - s_lse[t];
      ms[e] = (ez[e] / s_se[t]) * lp;
      t += dt; j += dj;
      const bool wrap = j >= Av; j = wrap ? j - A : j; t = wrap ? t + 1 : t;
    }
  }
  __syncthreads();
  ILT(5);
  // phase 5: lane 0 runs the recursion err_t = td_t + disc_t*c_t*err_{t+1} in reverse, inputs fetched sixteen steps at a time (the batch that
  // crosses t = 0 reads and writes the pad in front of the arrays); the other waves take the entropy sums meanwhile
  if (tid == 0) {
    float e = 0.0f;
    for (int t1 = T; t1 > 0; t1 -= 16) {
      float td[16], dc[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { td[u] = s_td[t1 - 1 - u]; dc[u] = s_dc[t1 - 1 - u]; }
#pragma unroll
      for (int u = 0; u < 16; ++u) { e = td[u] + dc[u] * e; s_err[t1 - 1 - u] = e; }
    }
  } else if (tid >= 64) {
    for (int t = tid - 64; t < T; t += IL_NT - 64) s_H[t] = -il_row_sum(ms + t * A, Av);
  }
  __syncthreads();
  ILT(6);
  // phase 6 (one thread per t): q values, advantages, the three loss terms, the per-step gradient coefficients
  for (int t = tid; t < T; t += IL_NT) {
    const float v = s_val[t], vn = s_val[t + 1], mask = s_mask[t], H = s_H[t];
    const float errors = (s_err[t] + v) - v;
    const float en = s_err[t + 1];
    const float qboot = t == T - 1 ? vn : ((en + vn) - vn) + vn;
    const float q = s_rew[t] + s_disc[t] * qboot;
    const float pgadv = s_cr[t] * (q - v);
    s_pg[t] = -s_lpa[t] * pgadv * mask;
    s_bl[t] = errors * errors * mask;
    s_en[t] = -H * mask;
    s_c1[t] = -pgadv * mask;
    s_c2[t] = ent_coef * mask;
    s_dv[t] = vf_coef * (-errors) * mask;
  }
  __syncthreads();
  ILT(7);
  // phase 7 (one thread per element of dzv's [T][32] rows): gradients; three lanes sum the loss terms in t order first
  if (tid < 3) {
    const float* src = tid == 0 ? s_pg : (tid == 1 ? s_bl : s_en);
    float acc = 0.0f;
    for (int t0 = 0; t0 < T; t0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[t0 + u];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const float n = acc + v[u]; acc = t0 + u < Tv ? n : acc; }
    }
    partials[b * 3 + tid] = acc;
  }
  for (int x = tid; x < T * 32; x += IL_NT) {
    const int t = x >> 5, j = x & 31;
    const int e = t * A + (j < Av ? j : 0);
    const float lp = (zs[e] - s_mx[t]) - s_lse[t], pj = ez[e] / s_se[t];
    const float d = s_c1[t] * ((j == s_act[t] ? 1.0f : 0.0f) - pj) + s_c2[t] * pj * (lp + s_H[t]);
    dzv[((size_t)t * Bm + b) * 32 + j] = j < Av ? d : (j == Av ? s_dv[t] : 0.0f);
  }
  // bootstrap row T: no gradient
  if (tid < 32) dzv[((size_t)T * Bm + b) * 32 + tid] = 0.0f;
  ILT(8);
}
// One THREAD per time step, serial over the actions: the form this path started from (and the order of operations the phases above keep);
// used when T x A is too large for the block's LDS.
__global__ __launch_bounds__(256) void impala_loss_column_kernel(const float* logits, const float* value, const float* mu_logits, const int32_t* actions,
                                   const float* rewards, const uint8_t* dones, const uint8_t* firststeps, int T1, int Bm, int A, int col0,
                                   int ld, float gamma, float vf_coef, float ent_coef, float* dzv, float* partials) {
  extern __shared__ float ish[];
  const int b = blockIdx.x, T = T1 - 1, tid = threadIdx.x, nt = blockDim.x;
  float* s_lpa = ish;            // [T]  log pi(a)
  float* s_cr = ish + T1;        // [T]  min(1, rho)
  float* s_td = ish + 2 * T1;    // [T]  cr * (r + disc*v' - v)
  float* s_dc = ish + 3 * T1;    // [T]  disc * cr
  float* s_err = ish + 4 * T1;   // [T]  err recursion
  float* s_pg = ish + 5 * T1;    // per-t loss terms, summed in t order by thread 0
  float* s_bl = ish + 6 * T1;
  float* s_en = ish + 7 * T1;
  // pass 1 (t-parallel): log pi(a), rho and the scan inputs
  for (int t = tid; t < T; t += nt) {
    const size_t r = (size_t)t * Bm + b, sidx = (size_t)t * ld + col0 + b;
    const int a = actions[sidx];
    const float* z = logits + r * A;
    const float* m = mu_logits + sidx * A;
    float mx = z[0], mm = m[0];
    for (int j = 1; j < A; ++j) { mx = z[j] > mx ? z[j] : mx; mm = m[j] > mm ? m[j] : mm; }
    float sz = 0.0f, sm = 0.0f;
    for (int j = 0; j < A; ++j) { sz += cbm_expf(z[j] - mx); sm += cbm_expf(m[j] - mm); }
    const float lpa = (z[a] - mx) - cbm_logf(sz);
    const float lma = (m[a] - mm) - cbm_logf(sm);
    const float rho = cbm_expf(lpa - lma);
    const float disc = (1.0f - (float)dones[sidx]) * gamma;
    const float cr = rho < 1.0f ? rho : 1.0f;
    s_lpa[t] = lpa; s_cr[t] = cr;
    s_td[t] = cr * ((rewards[sidx] + disc * value[r + Bm]) - value[r]);
    s_dc[t] = disc * cr;
  }
  __syncthreads();
  // pass 2 (serial, reverse): err_t = td_t + disc_t*c_t*err_{t+1}
  if (tid == 0) {
    float e = 0.0f;
    for (int t = T - 1; t >= 0; --t) { e = s_td[t] + s_dc[t] * e; s_err[t] = e; }
  }
  __syncthreads();
  // pass 3 (t-parallel): losses and gradients
  for (int t = tid; t < T; t += nt) {
    const size_t r = (size_t)t * Bm + b, sidx = (size_t)t * ld + col0 + b;
    const float disc = (1.0f - (float)dones[sidx]) * gamma;
    const float mask = 1.0f - (float)firststeps[sidx];
    const float lpa = s_lpa[t], cr = s_cr[t], err = s_err[t];
    const float errors = (err + value[r]) - value[r];
    float qboot;
    if (t == T - 1) qboot = value[r + Bm];
    else { const float en = s_err[t + 1]; qboot = ((en + value[r + Bm]) - value[r + Bm]) + value[r + Bm]; }
    const float q = rewards[sidx] + disc * qboot;
    const float pgadv = cr * (q - value[r]);
    s_pg[t] = -lpa * pgadv * mask;
    s_bl[t] = errors * errors * mask;
    const float* z = logits + r * A;
    const int a = actions[sidx];
    float mx = z[0];
    for (int j = 1; j < A; ++j) mx = z[j] > mx ? z[j] : mx;
    float se = 0.0f;
    for (int j = 0; j < A; ++j) se += cbm_expf(z[j] - mx);
    const float lse = cbm_logf(se);
    float H = 0.0f;
    for (int j = 0; j < A; ++j) { const float lp = (z[j] - mx) - lse; H += (cbm_expf(z[j] - mx) / se) * lp; }
    H = -H;
    s_en[t] = -H * mask;
    float* d = dzv + r * 32;
    for (int j = 0; j < A; ++j) {
      const float lp = (z[j] - mx) - lse, pj = cbm_expf(z[j] - mx) / se;
      d[j] = (-pgadv * mask) * ((j == a ? 1.0f : 0.0f) - pj) + ent_coef * mask * pj * (lp + H);
    }
    d[A] = vf_coef * (-errors) * mask;
    for (int j = A + 1; j < 32; ++j) d[j] = 0.0f;
  }
  // bootstrap row T: no gradient
  if (tid < 32) dzv[((size_t)T * Bm + b) * 32 + tid] = 0.0f;
  __syncthreads();
  if (tid == 0) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int t = 0; t < T; ++t) { a0 += s_pg[t]; a1 += s_bl[t]; a2 += s_en[t]; }
    partials[b * 3 + 0] = a0; partials[b * 3 + 1] = a1; partials[b * 3 + 2] = a2;
  }
}
__global__ void impala_stats_kernel(const float* partials, int Bm, float vf_coef, float ent_coef, float* stats4) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s[3] = {0, 0, 0};
  for (int b = 0; b < Bm; ++b) for (int q = 0; q < 3; ++q) s[q] += partials[b * 3 + q];
  stats4[1] = s[0]; stats4[2] = 0.5f * s[1]; stats4[3] = s[2];
  stats4[0] = stats4[1] + vf_coef * stats4[2] + ent_coef * stats4[3];
}
void launch_impala_loss(const float* logits, const float* value, const float* mu_logits, const int32_t* actions, const float* rewards,
                        const uint8_t* dones, const uint8_t* firststeps, int T1, int Bm, int A, int col0, int ld, float gamma,
                        float vf_coef, float ent_coef, float* dzv, float* partials, float* stats4, hipStream_t st) {
  const size_t lds = impala_loss_lds_bytes(T1, A);
  if (lds <= 64 * 1024)
    hipLaunchKernelGGL(impala_loss_kernel, dim3(Bm), dim3(IL_NT), lds, st, logits, value, mu_logits, actions, rewards, dones, firststeps, T1, Bm, A, col0, ld,
                       gamma, vf_coef, ent_coef, dzv, partials);
  else
    hipLaunchKernelGGL(impala_loss_column_kernel, dim3(Bm), dim3(T1 - 1 >= 192 ? 256 : 128), (size_t)8 * T1 * sizeof(float), st, logits, value, mu_logits,
                       actions, rewards, dones, firststeps, T1, Bm, A, col0, ld, gamma, vf_coef, ent_coef, dzv, partials);
  hipLaunchKernelGGL(impala_stats_kernel, dim3(1), dim3(64), 0, st, partials, Bm, vf_coef, ent_coef, stats4);
}

// ------------------------------------------------------------------------------------------
// optimizer: clip_by_global_norm + adam (ppo:492-500,629) / rmsprop_pytorch_style (impala:152-188).
// Pass 1: CBM_NORM_PARTS block partials of sum(g^2); pass 2: every block re-reduces the partials in the
// same fixed order (bit-identical norm everywhere, no atomics) and applies the elementwise update.
__global__ __launch_bounds__(256) void sqnorm_partials_kernel(const float* g, int64_t n, float grad_div, float* partials) {
  // A thread's elements (stride gridDim * 256) are requested sixteen at a time and added in ascending order: the order, hence the bits, of the
  // one-load-per-iteration loop (an element past n enters as +0: s + 0 * 0 == s), without a memory round trip per element.
  __shared__ float red[4];
  float s = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 16 * stride) {
    float x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int64_t i = i0 + u * stride; x[u] = i < n ? g[i] : 0.0f; }
    if (grad_div == 1.0f) {   // uniform: no division computed and discarded per element
#pragma unroll
      for (int u = 0; u < 16; ++u) s += x[u] * x[u];
    } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) { const float v = x[u] / grad_div; s += v * v; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ float norm_from_partials(const float* partials) {
  __shared__ float red[4];
  float s = 0.0f;
  for (int i = threadIdx.x; i < CBM_NORM_PARTS; i += 256) s += partials[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}
// Elementwise updates: one expression per element, shared by the scalar kernels and the four-elements-per-thread ones (16-byte loads / stores
// when the four vectors are 16-byte aligned — the context's own always are); -ffp-contract=off, so both forms give the same bits.
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, bool clip, float gn, float max_norm, float lr, float b1, float b2,
                                          float eps, float bc1, float bc2, float grad_div) {
  float gi = grad_div == 1.0f ? g : g / grad_div;
  if (clip) gi = (gi / gn) * max_norm;
  const float mi = (1.0f - b1) * gi + b1 * m;
  const float vi = (1.0f - b2) * (gi * gi) + b2 * v;
  m = mi; v = vi;
  const float u = (mi / bc1) / (sqrtf(vi / bc2) + eps);
  p = p + (-lr) * u;
}
__device__ __forceinline__ void rmsprop_elem(float& p, float g, float& nu, bool clip, float gn, float max_norm, float lr, float decay, float eps,
                                             float grad_div) {
  float gi = grad_div == 1.0f ? g : g / grad_div;
  if (clip) gi = (gi / gn) * max_norm;
  const float ni = (1.0f - decay) * (gi * gi) + decay * nu;
  nu = ni;
  p = p + (-lr) * (gi / (sqrtf(ni) + eps));
}
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float max_norm, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2, float grad_div,
                                                   const float* partials) {
  const float gn = norm_from_partials(partials);
  const bool clip = !(gn < max_norm);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    adam_elem(p[i], g[i], m[i], v[i], clip, gn, max_norm, lr, b1, b2, eps, bc1, bc2, grad_div);
}
__global__ __launch_bounds__(256) void adam_vec4_kernel(float* p, const float* g, float* m, float* v, int64_t n, float max_norm, float lr,
                                                        float b1, float b2, float eps, float bc1, float bc2, float grad_div,
                                                        const float* partials) {
  const float gn = norm_from_partials(partials);
  const bool clip = !(gn < max_norm);
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 G = reinterpret_cast<const float4*>(g)[i];
    float4 Pv = reinterpret_cast<float4*>(p)[i], M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    adam_elem(Pv.x, G.x, M.x, V.x, clip, gn, max_norm, lr, b1, b2, eps, bc1, bc2, grad_div);
    adam_elem(Pv.y, G.y, M.y, V.y, clip, gn, max_norm, lr, b1, b2, eps, bc1, bc2, grad_div);
    adam_elem(Pv.z, G.z, M.z, V.z, clip, gn, max_norm, lr, b1, b2, eps, bc1, bc2, grad_div);
    adam_elem(Pv.w, G.w, M.w, V.w, clip, gn, max_norm, lr, b1, b2, eps, bc1, bc2, grad_div);
    reinterpret_cast<float4*>(m)[i] = M; reinterpret_cast<float4*>(v)[i] = V; reinterpret_cast<float4*>(p)[i] = Pv;
  }
  const int64_t it = (n4 << 2) + threadIdx.x;   // the last n % 4 elements
  if (blockIdx.x == 0 && it < n) adam_elem(p[it], g[it], m[it], v[it], clip, gn, max_norm, lr, b1, b2, eps, bc1, bc2, grad_div);
}
__global__ __launch_bounds__(256) void rmsprop_kernel(float* p, const float* g, float* nu, int64_t n, float max_norm, float lr, float decay,
                                                      float eps, float grad_div, const float* partials) {
  const float gn = norm_from_partials(partials);
  const bool clip = !(gn < max_norm);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    rmsprop_elem(p[i], g[i], nu[i], clip, gn, max_norm, lr, decay, eps, grad_div);
}
__global__ __launch_bounds__(256) void rmsprop_vec4_kernel(float* p, const float* g, float* nu, int64_t n, float max_norm, float lr, float decay,
                                                           float eps, float grad_div, const float* partials) {
  const float gn = norm_from_partials(partials);
  const bool clip = !(gn < max_norm);
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 G = reinterpret_cast<const float4*>(g)[i];
    float4 Pv = reinterpret_cast<float4*>(p)[i], N = reinterpret_cast<float4*>(nu)[i];
    rmsprop_elem(Pv.x, G.x, N.x, clip, gn, max_norm, lr, decay, eps, grad_div);
    rmsprop_elem(Pv.y, G.y, N.y, clip, gn, max_norm, lr, decay, eps, grad_div);
    rmsprop_elem(Pv.z, G.z, N.z, clip, gn, max_norm, lr, decay, eps, grad_div);
    rmsprop_elem(Pv.w, G.w, N.w, clip, gn, max_norm, lr, decay, eps, grad_div);
    reinterpret_cast<float4*>(nu)[i] = N; reinterpret_cast<float4*>(p)[i] = Pv;
  }
  const int64_t it = (n4 << 2) + threadIdx.x;
  if (blockIdx.x == 0 && it < n) rmsprop_elem(p[it], g[it], nu[it], clip, gn, max_norm, lr, decay, eps, grad_div);
}
__global__ __launch_bounds__(256) void grad_accumulate_kernel(float* g, float* acc, int64_t n, float inv_steps_arg, float steps, int last, float grad_div) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = grad_div == 1.0f ? g[i] : g[i] / grad_div;
    const float a = (gi - acc[i]) / steps + acc[i];
    if (last) { g[i] = a; acc[i] = 0.0f; } else acc[i] = a;
  }
}
void launch_grad_accumulate(float* g, float* acc, int64_t n, int mini_step, bool last, float grad_div, hipStream_t st) {
  hipLaunchKernelGGL(grad_accumulate_kernel, dim3(1024), dim3(256), 0, st, g, acc, n, 0.0f, (float)(mini_step + 1), last ? 1 : 0, grad_div);
}
static bool opt_vec4_ok(const void* a, const void* b, const void* c, const void* d) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}
static int opt_vec4_blocks(int64_t n) { const int64_t b = ((n >> 2) + 255) / 256; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }
void launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float max_norm, float lr, float b1, float b2, float eps,
                 float bc1, float bc2, float grad_div, float* norm_partials, hipStream_t st) {
  hipLaunchKernelGGL(sqnorm_partials_kernel, dim3(CBM_NORM_PARTS), dim3(256), 0, st, g, n, grad_div, norm_partials);
  if (opt_vec4_ok(p, g, m, v))
    hipLaunchKernelGGL(adam_vec4_kernel, dim3(opt_vec4_blocks(n)), dim3(256), 0, st, p, g, m, v, n, max_norm, lr, b1, b2, eps, bc1, bc2, grad_div, norm_partials);
  else
    hipLaunchKernelGGL(adam_kernel, dim3(1024), dim3(256), 0, st, p, g, m, v, n, max_norm, lr, b1, b2, eps, bc1, bc2, grad_div, norm_partials);
}
void launch_rmsprop(float* p, const float* g, float* nu, int64_t n, float max_norm, float lr, float decay, float eps, float grad_div,
                    float* norm_partials, hipStream_t st) {
  hipLaunchKernelGGL(sqnorm_partials_kernel, dim3(CBM_NORM_PARTS), dim3(256), 0, st, g, n, grad_div, norm_partials);
  if (opt_vec4_ok(p, g, nu, nu))
    hipLaunchKernelGGL(rmsprop_vec4_kernel, dim3(opt_vec4_blocks(n)), dim3(256), 0, st, p, g, nu, n, max_norm, lr, decay, eps, grad_div, norm_partials);
  else
    hipLaunchKernelGGL(rmsprop_kernel, dim3(1024), dim3(256), 0, st, p, g, nu, n, max_norm, lr, decay, eps, grad_div, norm_partials);
}
