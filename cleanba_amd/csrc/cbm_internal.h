// cbm_internal.h — internal C++ interfaces between the translation units of libcleanba_mi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/cleanba_mi.h"
#include "../../include/cbm_math.h"

// A uniform loop bound (number of actions, rollout length) copied into a VGPR the compiler cannot see through: comparisons against it become
// per-lane selects.  Against the SGPR itself an `if (j < A)` inside an unrolled loop is a scalar branch per iteration — in the small latency-bound
// kernels (loss heads, heads' input gradient) those branches, each with its own wait in front, were most of the run time (profiles/NOTES_r03_r04.md, round 4).
static __device__ __forceinline__ int cbm_opaque_vgpr(int x) { asm volatile("" : "+v"(x)); return x; }

#define CBM_FRAME 28224  // 4*84*84 uint8

// global -> LDS copy by the load unit (global_load_lds_dwordx4: 16 bytes per lane to wave-uniform LDS byte address + lane * 16), issued from inline
// asm so that hipcc does NOT know it writes LDS.  Through __builtin_amdgcn_global_load_lds the compiler tracks the copy as a pending LDS store and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read it cannot tell apart from it — whether it can depends on memory-operand bookkeeping that changes
// with unrelated edits (a second __shared__ object, two reads merged into a ds_read2st64): in round 5's frame-resident weight gradients that wait sat
// between "issue the copy of frame s+1" and "multiply frame s", i.e. the double buffer never overlapped anything (tools/isa_audit.py lists such
// waits).  Hidden, the copy is ordered ONLY by the kernel's own `s_waitcnt vmcnt(N)` + barrier; the "memory" clobber keeps the compiler's LDS
// accesses on their side of the statement, and a copy the compiler does not count only makes its own vmcnt waits for ordinary loads more
// conservative (the counter retires in order).  M0 is saved and restored inside the statement (cdna_hip_programming.md section 5.7).
static __device__ __forceinline__ uint32_t cbm_lds_addr(const void* lds_ptr) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)lds_ptr;
}
static __device__ __forceinline__ void cbm_glds16_hidden(const void* g_lane, uint32_t lds_byte_addr_wave) {
  uint32_t keep;
  const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte_addr_wave);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g_lane), "s"(dst) : "memory");
}
// write-through (sc1) 4-byte store: the value leaves this XCD's L2 at once — what a workgroup of the same launch on another XCD can read with an sc1 load
#ifndef AF_ABL   // timing builds only (tools/variants.sh): 1 = plain stores, 2 = plain loads, 4 = no waiting — wrong results, informative times
#define AF_ABL 0
#endif
static __device__ __forceinline__ void cbm_store_wt(float* p, float v) {
  if (AF_ABL & 1) { *p = v; return; }
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- parameter layout (flax shapes): Nature-CNN ------------------------------
struct NatureLayout {
  int A;
  int kind = 0, hid = 512, flat = 3136;  // CBM_NET_NATURE: 3136 -> 512;  CBM_NET_IMPALA_RESNET: 3872 -> 256
  int64_t w[6], b[6];  // 0 conv1, 1 conv2, 2 conv3, 3 dense, 4 actor, 5 critic  (ResNet: only 3..5)
  int64_t rcw[3][5], rcb[3][5];  // ResNet: per ConvSequence {Conv_0, RB0.Conv_0, RB0.Conv_1, RB1.Conv_0, RB1.Conv_1}
  int64_t total;
};
NatureLayout nature_layout(int A);
NatureLayout net_layout(int kind, int A, int hid = 0);   // hid: width of the hidden layer (0 = the network's default: 512 / 256)

// ---- optional per-kernel HIP-event timing (bench.py roofline): events bracket every launch of the
// selected kernel id on the stream it is launched on.
enum { K_CONV1_FWD = 0, K_CONV2_FWD, K_CONV3_FWD, K_DENSE_FWD, K_HEADS_WGRAD, K_DENSE_DGRAD, K_DENSE_WGRAD, K_CONV3_DGRAD,
       K_CONV3_WGRAD, K_CONV2_DGRAD, K_CONV2_WGRAD, K_CONV1_WGRAD, K_NUM };
#define CBM_PROF_MAX 16384
#define CBM_PROF_PAUSE (-3) // stop recording but keep the recorded launches readable
#define CBM_PROF_ALL (-2)   // time every launch of every id (bench.py picks the dominant kernel by measured time)
struct CbmProf {
  int sel = -1;
  int n = 0;
  int8_t kid[CBM_PROF_MAX];
  hipEvent_t ev[2 * CBM_PROF_MAX];
  bool created = false;
  // what was launched for each id the last time it was timed: kernel symbol + problem functor (static strings; cbm_profile_kernel_name)
  const char* kernel[16] = {};
  const char* functor[16] = {};
};

// ---- workspace for running the network on up to maxB frames ----------------------------
// The fused PPO heads leave their block partials of the loss statistics; the sum rides in the backward pass's reduction launch
// (gemm_layers.hip: RedBatch::add_stats), or in its own launch when no backward pass follows (flush_pending_stats).
struct PendingStats { const float* partials = nullptr; float* stats5 = nullptr; int nblk = 0, N = 0; float ent_coef = 0.0f, vf_coef = 0.0f; };
struct NatureWs {
  PendingStats pending_stats;
  CbmProf* prof = nullptr;
  int maxB = 0;
  bool with_grad = false;
  hipEvent_t tail_ev = nullptr;   // when set: recorded by the backward pass once the gradients of dense + heads (the flat tail [w[3], total)) are final
  int bwd_split = 0;       // cbm_config.backward_split: 0 fp32 MFMA, 2 / 3 split-bf16 backward GEMMs (igemm_split_kernel)
  bool conv1_exact_fwd = false, conv1_exact_wgrad = false;   // learner-size conv1 forward / weight gradient as exact uint8 x 3-term-bf16 products on the bf16 matrix cores (conv1.hip); false = fp32 MFMA chains
  bool bf16_fwd = false;   // cbm_config.forward_bf16: conv2/conv3/dense forward on bf16 MFMA (Nature-CNN)
  bool skip_heads = false; // set around a forward / backward pair whose heads forward, PPO loss and heads dgrad run as launch_ppo_heads_fused
  float *act1 = nullptr, *act2 = nullptr, *act3 = nullptr, *hid = nullptr;
  float *logits = nullptr, *value = nullptr;
  float* dense_part = nullptr;  // [ksplit][maxB][512] when maxB is small
  int dense_part_ksplit = 0;
  // backward
  float *dzv = nullptr, *dhid = nullptr, *dact3pad = nullptr, *dact2pad = nullptr, *dact1 = nullptr;
  float *wg_part = nullptr, *bias_part = nullptr;
  // ReLU masks as bits, written by the forward epilogues (one 32-bit word per 32 channels of an output row) and read by the dgrad
  // epilogues instead of the fp32 activations: act1 [B*400] words, act2 [B*81][2], act3 [B*49][2]
  uint32_t *mask1 = nullptr, *mask2 = nullptr, *mask3 = nullptr;
  int32_t* c3_order = nullptr;   // conv3 dgrad tile order (position-major tiles, heavy taps first at the end of each XCD's run)
  int c3_order_S = -1;
  int64_t wg_part_floats = 0, bias_part_floats = 0;
  // IMPALA-ResNet activations (resnet_layers.inc): per sequence {c0, p, b0y1, b0out, b1y1, b1out}, pool arg-max, grad ping-pong
  int kind = 0;
  float* rn_t[3][6] = {};
  uint8_t* rn_pidx[3] = {};
  void* rn_m[3][5] = {};   // relu bit masks (value > 0, C bits per position) of rn_t[s][1..4], written by the forward at learner sizes (learner workspaces only)
  float* rn_tr[3][2] = {};   // relu'd copies of rn_t[s][1] / rn_t[s][3] (the inputs of the residual blocks' first convs), learner workspaces: see rn_seq_forward
  float* rn_g[2] = {};
  float* rn_wT = nullptr;  // flipped/transposed conv weights for the dgrad convs (rebuilt per backward)
  // dataflow actor step (gemm_layers.hip actor_fused_kernel): per-frame arrival counters of act1 / act2 / act3 (3 x maxB words, monotonic: never reset),
  // the number of fused launches so far, and a page-locked word a block sets when it gives up waiting for its producers
  uint32_t* af_cnt = nullptr;
  uint32_t af_epoch = 0;
  uint32_t* af_err_host = nullptr;   // host address of the mapped word
  uint32_t* af_err_dev = nullptr;    // its device address
};
int nature_ws_alloc(NatureWs& ws, int maxB, bool with_grad, int dense_ksplit_small, int kind = 0);
void nature_ws_free(NatureWs& ws);

// get_action_and_value's sampling step handed to the forward pass so that an actor step can end in ONE launch (split-K reduce of the dense
// layer + heads + Gumbel arg-max + log-softmax): same outputs as launch_sample on ws.logits / ws.value.
struct ActorSample {
  uint32_t sk0, sk1;          // subkey words (jax.random.split(key)[1])
  int32_t* actions;           // [B]
  float* logprobs;            // [B] or null (IMPALA)
  float* value_out;           // [B] or null
  float* logits_out;          // [B][A] or null (PPO)
  // optional: the device env's step with the sampled action, done by the launch that sampled it (nature_forward returns 2 then)
  uint32_t env_seed; int32_t env_max_steps; cbm_env_state* env_st; const uint8_t* env_obs_prev; uint8_t* env_obs_next; float* env_reward;
  uint8_t* env_done_next; uint8_t* env_firststep_next;   // env_obs_next == nullptr: no env step
};
// forward: obs[idx[b]] (idx may be null) -> ws.logits [B,A], ws.value [B]; activations kept in ws.  With `sample` non-null the call MAY
// also do the sampling (returns true then; ws.logits / ws.value / ws.hid are not written); false = the caller launches launch_sample.
int nature_forward(const NatureLayout& L, const float* P, const uint8_t* obs, const int32_t* idx, int B,
                    int dense_ksplit, NatureWs& ws, hipStream_t st, const ActorSample* sample = nullptr);
// backward from ws.dzv ([B][32]: dlogits | dvalue | 0) -> grads (flat, same layout as params).
void nature_backward(const NatureLayout& L, const float* P, const uint8_t* obs, const int32_t* idx, int B,
                     NatureWs& ws, float* grads, hipStream_t st);

// frame-resident conv1 kernels (conv1.hip)
void launch_conv1_fwd_frames(const uint8_t* obs, const int32_t* idx, const float* W, const float* bias, float* out, uint32_t* mask, int S,
                             hipStream_t st, bool exact = false);
int conv1_wgrad_frames_splits(int S);
int conv1_wgrad_frames_splits_bound(int maxS);   // >= conv1_wgrad_frames_splits(S) for every S <= maxS (the split count is not monotone in S)
void launch_conv1_wgrad_frames(const uint8_t* obs, const int32_t* idx, const float* dy, float* part, float* bpart, int S, hipStream_t st,
                               bool split = false, bool exact = false);

// frame-resident conv2 / conv3 weight gradients (wgrad_frames.hip): partials part[z][(kh,kw,ci)][co], bpart[z][co]
int conv2_wgrad_frames_splits(int S);
int conv2_wgrad_frames_splits_bound(int maxS);
void launch_conv2_wgrad_frames(const float* act1, const float* dypad, float* part, float* bpart, int S, hipStream_t st);
int conv3_wgrad_frames_splits(int S);
int conv3_wgrad_frames_splits_bound(int maxS);
void launch_conv3_wgrad_frames(const float* act2, const float* dypad, float* part, float* bpart, int S, hipStream_t st);

// load-unit fed weight gradient of the dense layer (dense_wgrad.hip): partials part[z][X][Y], bpart[z][Y]; slices = 0: batch not handled (use the igemm)
int dense_wgrad_dma_slices(int F);
// conv_regw.hip: learner-size conv2 / conv3 forward with the weights in registers (bit-identical to ConvFwd on igemm_kernel)
void launch_conv3_fwd_regw(const float* in, const float* W, const float* bias, float* out, uint32_t* mask, int B, hipStream_t st);
void launch_conv2_fwd_regw(const float* in, const float* W, const float* bias, float* out, uint32_t* mask, int B, hipStream_t st);
void launch_dense_wgrad_dma(const float* act, const float* dhid, float* part, float* bpart, int F, int X, int Y, int nz, hipStream_t st);

// ---- pointwise / scan kernels -----------------------------------------------------------
void launch_sample(const float* logits, int B, int A, uint32_t sk0, uint32_t sk1, int32_t* actions, float* logprobs,
                   const float* value_in, float* value_out, float* logits_out, hipStream_t st);
void launch_gae(const float* rewards, const float* values, const uint8_t* dones, const float* next_value,
                const uint8_t* next_done, int T, int B, float gamma, float lambda, float* adv, float* target,
                hipStream_t st);
void launch_advnorm(float* adv, int T, int B, int groups, hipStream_t st);
void launch_gae_async(const int32_t* env_ids, const float* rewards, const float* values, const uint8_t* dones, int R, int B, int num_envs,
                      float gamma, float lambda, float* adv, float* target, hipStream_t st);
void launch_vtrace(const float* v_tm1, const float* v_t, const float* r_t, const float* disc_t, const float* rho_tm1, int T, int B, float* errors,
                   float* pg_adv, float* q_est, hipStream_t st);
void launch_mb_advnorm(const float* adv, const int32_t* idx, int n, float* out, hipStream_t st);
// perm = jax.random.permutation(key, n): host does the key splits, device the bits + stable sorts.
size_t permutation_scratch_u64(int n);   // uint64 words of scratch launch_permutation needs for n elements
void launch_permutation(const uint32_t key[2], int n, int32_t* perm, int32_t* tmp, uint64_t* scratch, hipStream_t st);
// every epoch's permutation of one update at once (perms / tmps [ne][n], scratch of permutation_batch_scratch_u64(n, ne) words); false = not
// taken (one epoch, or more than CBM_PERM_BATCH_MAX epoch x round jobs): the caller permutes epoch by epoch
#define CBM_PERM_BATCH_MAX 16
size_t permutation_batch_scratch_u64(int n, int ne);
bool permutation_batch_ok(int n, int ne);   // whether launch_permutations_batch takes this (n, ne)
bool launch_permutations_batch(const uint32_t (*epoch_keys)[2], int ne, int n, int32_t* perms, int32_t* tmps, uint64_t* scratch, hipStream_t st);
void launch_ppo_loss(const float* logits, const float* value, int N, int A, const int32_t* idx, const int32_t* actions,
                     const float* old_logprob, const float* adv, const float* target, float clip_coef, float ent_coef,
                     float vf_coef, float* dzv, float* partials, float* stats5, hipStream_t st);
// the block partials of ppo_loss_kernel / the fused heads kernel -> the five statistics of ppo:649-653
void launch_ppo_stats(const float* partials, int nblk, int N, float ent_coef, float vf_coef, float* stats5, hipStream_t st);
void flush_pending_stats(NatureWs& ws, hipStream_t st);   // statistics left by launch_ppo_heads_fused when no backward pass consumed them
// ppo_stats_kernel's arithmetic for one wave (shared with the reduction launch's statistics job: the same order, the same bits)
__device__ __forceinline__ void ppo_stats_wave(const float* partials, int nblk, int N, float ent_coef, float vf_coef, float* stats5) {
  const int l = threadIdx.x;
  float s[4] = {0, 0, 0, 0};
  for (int b = l; b < nblk; b += 64) for (int q = 0; q < 4; ++q) s[q] += partials[b * 4 + q];
  for (int q = 0; q < 4; ++q) for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_down(s[q], o, 64);
  if (l != 0) return;
  const float n = (float)N;
  const float pg = s[0] / n, v = 0.5f * (s[1] / n), e = s[2] / n, kl = s[3] / n;
  stats5[0] = pg - ent_coef * e + v * vf_coef;
  stats5[1] = pg; stats5[2] = v; stats5[3] = e; stats5[4] = kl;
}
// heads forward + PPO loss + heads input gradient of a minibatch in one launch (gemm_layers.hip): hid [B][HD] -> logits / value (ws), dzv [B][32],
// dhid [B][HD], the five statistics.  Same logits bits as launch_heads_fwd, same loss arithmetic as launch_ppo_loss.
bool ppo_heads_fusable(const NatureLayout& L);
void launch_ppo_heads_fused(const NatureLayout& L, const float* P, NatureWs& ws, int B, const int32_t* idx, const int32_t* actions,
                            const float* old_logprob, const float* adv, const float* target, float clip_coef, float ent_coef, float vf_coef,
                            float* partials, float* stats5, hipStream_t st);
void launch_impala_loss(const float* logits, const float* value, const float* mu_logits, const int32_t* actions,
                        const float* rewards, const uint8_t* dones, const uint8_t* firststeps, int T1, int Bm, int A,
                        int col0, int ld, float gamma, float vf_coef, float ent_coef, float* dzv, float* partials,
                        float* stats4, hipStream_t st);
void launch_grad_accumulate(float* g, float* acc, int64_t n, int mini_step, bool last, float grad_div, hipStream_t st);
void launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float max_norm, float lr, float b1, float b2,
                 float eps, float bc1, float bc2, float grad_div, float* norm_partials, hipStream_t st);
void launch_rmsprop(float* p, const float* g, float* nu, int64_t n, float max_norm, float lr, float decay, float eps,
                    float grad_div, float* norm_partials, hipStream_t st);
#define CBM_NORM_PARTS 512

// ---- synthetic env ------------------------------------------------------------------------
void launch_env_reset(uint32_t seed, int E, int atari57_mix, cbm_env_state* st_dev, uint8_t* obs, int64_t obs_stride, uint8_t* done,
                      uint8_t* firststep, hipStream_t st);
void launch_env_step(uint32_t seed, int E, int max_episode_steps, const int32_t* actions, cbm_env_state* st_dev,
                     const uint8_t* obs_prev, uint8_t* obs_next, float* reward, uint8_t* done_next,
                     uint8_t* firststep_next, hipStream_t st);
void launch_env_stats(const cbm_env_state* st_dev, int E, float* out2, hipStream_t st);

// error plumbing
void cbm_set_error(const char* fmt, ...);
// A launch helper found an internal invariant violated (a ring / partial-region geometry that does not cover the batch it was handed): the
// launch is skipped, the message is kept until the C-ABI entry point that drove the pass returns -1 with it (cbm_launch_check, which then clears it) —
// the host gets a cbm error instead of an abort().
void cbm_launch_fail(const char* fmt, ...);
int cbm_launch_check(void);   // 0 = no launch helper has failed; -1 = one has, cbm_last_error() carries its message
#define CBM_HIP(call)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      cbm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                           \
    }                                                                                      \
  } while (0)
