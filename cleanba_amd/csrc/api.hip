// api.hip — the C ABI of libcleanba_mi.so (include/cleanba_mi.h): context, HBM rollout ring,
// actor / learner orchestration on HIP streams, and the pure-function entry points used by tests.
//
// Hand-off design (replaces the two queue.Queue(maxsize=1) per actor thread, ppo:662-686):
//   - rollouts live in an HBM ring of `ring_depth` entries; every field is [T+1][B_dev] (t-major,
//     env columns of slot s at [s*E,(s+1)*E)), exactly the hstack/flatten order of ppo:587,601.
//   - producer (actor slot) and consumer (learner) exchange only sequence numbers on the host
//     (committed[s], updates_done: lock-free atomics, sleepers on a futex word) and HIP events on the device (ready / consumed / params_ready);
//     the GPU never waits for the host and the host never copies rollout data.
//   - parameters are published by the learner into a 3-deep versioned buffer; an actor rollout
//     `u` reads version max(0,u-2) with --concurrency (the `update != 2` skew, ppo:287-304) or
//     u-1 without.
#include "cbm_ctx.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <mutex>
#include <condition_variable>
#include <vector>

static thread_local char g_err[512] = "";
void cbm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* cbm_last_error(void) { return g_err; }
// A launch helper's failure is recorded PER THREAD: the helpers run inside the C-ABI call that drives the pass, on the caller's thread, and that call ends
// in cbm_launch_check() on the same thread.  (A process-wide flag — rounds 4-5 — let thread B's entry point consume and report a launch that thread
// A's pass had skipped, while A's own call returned 0 although its kernel never ran)  Reported once, then cleared: one context's geometry
// failure does not make later calls on that thread fail with a stale message.
static thread_local char g_launch_err[512] = "";
static thread_local bool g_launch_failed = false;
void cbm_launch_fail(const char* fmt, ...) {
  if (g_launch_failed) return;   // keep the first one
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_launch_err, sizeof(g_launch_err), fmt, ap);
  va_end(ap);
  g_launch_failed = true;
}
int cbm_launch_check(void) {
  if (!g_launch_failed) return 0;
  cbm_set_error("%s", g_launch_err);
  g_launch_failed = false;
  return -1;
}
extern "C" const char* cbm_build_info(void) { return "cleanba-mi gfx950 f32-mfma abi=2 built " __DATE__ " " __TIME__; }

static bool is_ppo(const cbm_ctx* c) { return c->cfg.algo == CBM_ALGO_PPO; }

extern "C" int cbm_default_config(int32_t algo, cbm_config* cfg) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->abi_version = CBM_ABI_VERSION;
  cfg->network = CBM_NET_NATURE;
  cfg->algo = algo;
  cfg->num_actions = 18;
  cfg->local_num_envs = 64;
  cfg->num_actor_slots = 2;
  cfg->num_minibatches = 4;
  cfg->ring_depth = 3;
  cfg->gamma = 0.99f;
  cfg->gae_lambda = 0.95f;
  cfg->ent_coef = 0.01f;
  cfg->vf_coef = 0.5f;
  cfg->adam_b1 = 0.9f; cfg->adam_b2 = 0.999f; cfg->adam_eps = 1e-5f;
  cfg->rms_decay = 0.99f; cfg->rms_eps = 0.01f;
  cfg->actor_dense_ksplit = 14;
  cfg->num_channels = 3; cfg->channels[0] = 16; cfg->channels[1] = 32; cfg->channels[2] = 32;   // ppo:92
  cfg->num_hiddens = 1; cfg->hiddens[0] = 256;                                                  // ppo:94
  if (algo == CBM_ALGO_PPO) {
    cfg->num_steps = 128; cfg->update_epochs = 4; cfg->norm_adv = 1; cfg->clip_coef = 0.1f; cfg->max_grad_norm = 0.5f;
  } else {
    cfg->num_steps = 20; cfg->update_epochs = 1; cfg->norm_adv = 0; cfg->clip_coef = 0.0f; cfg->max_grad_norm = 40.0f;
  }
  return 0;
}

extern "C" int32_t cbm_config_size(void) { return (int32_t)sizeof(cbm_config); }

extern "C" int64_t cbm_param_count_hidden(int32_t network, int32_t num_actions, int32_t hidden) {
  if (network != CBM_NET_NATURE && network != CBM_NET_IMPALA_RESNET) return -1;
  return net_layout(network, num_actions, network == CBM_NET_IMPALA_RESNET ? hidden : 0).total;
}
extern "C" int64_t cbm_param_count(int32_t network, int32_t num_actions) {
  if (network != CBM_NET_NATURE && network != CBM_NET_IMPALA_RESNET) return -1;
  return net_layout(network, num_actions).total;
}

template <class Tp>
static int dalloc(Tp** p, size_t n) {
  if (hipMalloc((void**)p, n * sizeof(Tp)) != hipSuccess) { cbm_set_error("hipMalloc(%zu bytes) failed", n * sizeof(Tp)); return -1; }
  return 0;
}

static int ctx_create_impl(const cbm_config* cfg, cbm_ctx** out, cbm_ctx** partial) {
  if (!cfg || cfg->abi_version != CBM_ABI_VERSION) { cbm_set_error("bad config / abi version"); return -1; }
  if (cfg->network != CBM_NET_NATURE && cfg->network != CBM_NET_IMPALA_RESNET) { cbm_set_error("unknown network kind %d", cfg->network); return -2; }
  if (cfg->num_actor_slots < 1 || cfg->num_actor_slots > MAX_SLOTS || cfg->ring_depth < 2 || cfg->ring_depth > MAX_RING) {
    cbm_set_error("num_actor_slots in [1,%d], ring_depth in [2,%d]", MAX_SLOTS, MAX_RING); return -1;
  }
  if (cfg->num_actions < 2 || cfg->num_actions > 28) { cbm_set_error("num_actions must be in [2,28]"); return -1; }
  if (cfg->algo == CBM_ALGO_IMPALA && cfg->num_steps + 1 > 2000) { cbm_set_error("IMPALA num_steps must be <= 1999 (the V-trace kernel keeps 8 floats per step in LDS)"); return -1; }
  if (cfg->backward_split != 0 && cfg->backward_split != 2 && cfg->backward_split != 3) { cbm_set_error("backward_split must be 0, 2 or 3"); return -1; }
  if (cfg->conv1_fp32_chain < 0 || cfg->conv1_fp32_chain > 3) { cbm_set_error("conv1_fp32_chain must be 0..3 (bit 0: forward, bit 1: weight gradient on the fp32 chain)"); return -1; }
  if (cfg->backward_split && cfg->network != CBM_NET_NATURE) { cbm_set_error("backward_split is built for the Nature-CNN torso only"); return -1; }
  if (cfg->network == CBM_NET_IMPALA_RESNET) {
    // the slab-convolution geometries are compiled for the reference's channel widths (ppo:92-93); the hidden layer (ppo:94) may be any multiple of
    // 64 up to 512 (the dense / heads kernels take its width at run time), one layer
    if (!(cfg->num_channels == 3 && cfg->channels[0] == 16 && cfg->channels[1] == 32 && cfg->channels[2] == 32)) {
      cbm_set_error("--channels: only the reference default [16, 32, 32] is built into the HIP ResNet torso");
      return -1;
    }
    if (!(cfg->num_hiddens == 1 && cfg->hiddens[0] >= 64 && cfg->hiddens[0] <= 512 && cfg->hiddens[0] % 64 == 0)) {
      cbm_set_error("--hiddens: the HIP ResNet torso takes ONE hidden layer of 64, 128, ... 512 units (reference default [256])");
      return -1;
    }
  }
  if (cfg->forward_bf16 && cfg->network != CBM_NET_NATURE) { cbm_set_error("forward_bf16 is built for the Nature-CNN torso only"); return -1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { cbm_set_error("no HIP device visible: libcleanba_mi needs an MI355X"); return -3; }
  CBM_HIP(hipSetDevice(cfg->device));
  cbm_ctx* c = new cbm_ctx();
  *partial = c;   // the wrapper releases whatever was allocated if anything below fails
  c->cfg = *cfg;
  c->A = cfg->num_actions; c->E = cfg->local_num_envs; c->S = cfg->num_actor_slots;
  c->T = cfg->num_steps;
  if (cfg->async_batch_size > 0) {
    // legacy async mode (naturecnn:65-66,104): a rollout is num_steps*async_update rows of async_batch_size samples, one actor thread
    if (cfg->algo != CBM_ALGO_PPO) { cbm_set_error("async_batch_size is a PPO option (the IMPALA script already uses recv/send with all envs)"); return -1; }
    if (cfg->num_actor_slots != 1) { cbm_set_error("async_batch_size supports one actor slot (naturecnn:105)"); return -1; }
    if (cfg->local_num_envs % cfg->async_batch_size) { cbm_set_error("local_num_envs must be a multiple of async_batch_size"); return -1; }
    c->asyncB = cfg->async_batch_size; c->NE = cfg->local_num_envs;
    c->E = c->asyncB; c->T = cfg->num_steps * (c->NE / c->asyncB);
  }
  c->Bdev = c->E * c->S; c->T1 = c->T + 1;
  c->nmb = cfg->num_minibatches; c->epochs = is_ppo(c) ? cfg->update_epochs : 1;
  c->accum = cfg->grad_accum_steps > 1 ? cfg->grad_accum_steps : 1;
  c->nmicro = c->nmb * c->accum;   // micro-batches per epoch (ppo:607, impala:627)
  if (c->Bdev > 1024) { cbm_set_error("local_num_envs*slots must be <= 1024 per GPU"); return -1; }
  if (!c->asyncB && c->Bdev % c->nmb) { cbm_set_error("local_num_envs*slots must be divisible by num_minibatches (ppo:416-418)"); return -1; }
  if (is_ppo(c) ? (c->T * c->Bdev) % c->nmicro : c->Bdev % c->nmicro) {
    cbm_set_error("the batch does not split into num_minibatches*gradient_accumulation_steps = %d micro-batches", c->nmicro); return -1;
  }
  c->MB = is_ppo(c) ? (c->T * c->Bdev) / c->nmicro : c->T1 * (c->Bdev / c->nmicro);
  c->L = net_layout(cfg->network, c->A, cfg->network == CBM_NET_IMPALA_RESNET ? cfg->hiddens[0] : 0);
  if (c->L.flat % cfg->actor_dense_ksplit || (c->L.flat / cfg->actor_dense_ksplit) % 32) {
    cbm_set_error("actor_dense_ksplit must cut the %d-wide flatten into multiples of 32 (Nature: 14, ResNet: 11)", c->L.flat); return -1;
  }
  c->P = c->L.total;
  const size_t P = (size_t)c->P, B = (size_t)c->Bdev, T1 = (size_t)c->T1;
  c->stat_rows = c->epochs * c->nmicro;
  // The two export windows (cbm_ctx.h): every buffer a peer process may map is carved from one of them at 4 KB granularity, so a peer maps ONE
  // allocation per context and purpose (never one of the runtime's sub-allocated fragments) and addresses fields by offset (cbm_ipc_window_offset).
  {
    auto carve = [](size_t& off, size_t bytes) { const size_t o = off; off = (off + bytes + 4095) & ~(size_t)4095; return o; };
    struct RingOff { size_t obs, actions, logprobs, values, rewards, logits, dones, firststeps, env_ids; } ro[MAX_RING];
    size_t w0 = 0, ap[NPV];
    for (int i = 0; i < NPV; ++i) ap[i] = carve(w0, P * 4);
    for (int r = 0; r < cfg->ring_depth; ++r) {
      ro[r].obs = carve(w0, T1 * B * CBM_FRAME); ro[r].actions = carve(w0, T1 * B * 4); ro[r].logprobs = carve(w0, T1 * B * 4);
      ro[r].values = carve(w0, T1 * B * 4); ro[r].rewards = carve(w0, T1 * B * 4); ro[r].logits = carve(w0, T1 * B * c->A * 4);
      ro[r].dones = carve(w0, T1 * B); ro[r].firststeps = carve(w0, T1 * B); ro[r].env_ids = carve(w0, T1 * B * 4);
    }
    size_t w1 = 0;
    const size_t o_grads = carve(w1, P * 4), o_stats = carve(w1, (size_t)c->stat_rows * 8 * 4), o_scr = carve(w1, CBM_COMM_SCRATCH * 8);
    const size_t min_win = (size_t)4 << 20;
    c->win_bytes[0] = std::max(w0, min_win);
    c->win_bytes[1] = std::max(w1, min_win);
    // window 1 is what the native all-reduce's kernels read and write in PEER memory while they run: across devices that needs fine-grained
    // memory (coarse-grained memory is only coherent at kernel boundaries).  CBM_NATIVE_FINEGRAINED=1 / 0 forces it; default: fine-grained when the
    // host asked for the native backend (CBM_COMM=native) without pinning every rank to one device (CBM_FORCE_DEVICE).
    const char* fg = getenv("CBM_NATIVE_FINEGRAINED");
    const char* be = getenv("CBM_COMM");
    c->win_fine[1] = fg ? atoi(fg) != 0 : (be && !strcasecmp(be, "native") && !getenv("CBM_FORCE_DEVICE") && ndev > 1);
    for (int w = 0; w < CBM_WINDOWS; ++w) {
      void* p = nullptr;
      const hipError_t e = c->win_fine[w] ? hipExtMallocWithFlags(&p, c->win_bytes[w], hipDeviceMallocFinegrained) : hipMalloc(&p, c->win_bytes[w]);
      if (e != hipSuccess) { cbm_set_error("export window %d: %s of %zu bytes failed: %s", w, c->win_fine[w] ? "fine-grained allocation" : "hipMalloc", c->win_bytes[w], hipGetErrorString(e)); return -1; }
      c->win[w] = (uint8_t*)p;
    }
    CBM_HIP(hipMemset(c->win[1], 0, c->win_bytes[1]));
    c->grads = (float*)(c->win[1] + o_grads); c->stats_dev = (float*)(c->win[1] + o_stats); c->comm_scratch = (double*)(c->win[1] + o_scr);
    for (int i = 0; i < NPV; ++i) c->actor_params[i] = (float*)(c->win[0] + ap[i]);
    for (int r = 0; r < cfg->ring_depth; ++r) {
      RingEntry& R = c->ring[r];
      uint8_t* const w = c->win[0];
      R.obs = w + ro[r].obs; R.actions = (int32_t*)(w + ro[r].actions); R.logprobs = (float*)(w + ro[r].logprobs); R.values = (float*)(w + ro[r].values);
      R.rewards = (float*)(w + ro[r].rewards); R.logits = (float*)(w + ro[r].logits); R.dones = w + ro[r].dones; R.firststeps = w + ro[r].firststeps;
      R.env_ids = (int32_t*)(w + ro[r].env_ids);
    }
  }
  if (dalloc(&c->params, P) || dalloc(&c->opt_m, P) || dalloc(&c->opt_v, P)) return -1;
  hipMemset(c->params, 0, P * 4); hipMemset(c->opt_m, 0, P * 4); hipMemset(c->opt_v, 0, P * 4);
  for (int i = 0; i < NPV; ++i) { hipMemset(c->actor_params[i], 0, P * 4); CBM_HIP(hipEventCreateWithFlags(&c->params_ready[i], hipEventDisableTiming)); }
  for (int r = 0; r < cfg->ring_depth; ++r) {
    RingEntry& R = c->ring[r];
    hipMemset(R.env_ids, 0, T1 * B * 4); hipMemset(R.actions, 0, T1 * B * 4); hipMemset(R.logprobs, 0, T1 * B * 4); hipMemset(R.values, 0, T1 * B * 4);
    hipMemset(R.logits, 0, T1 * B * c->A * 4);   // rows a rollout never writes (PPO row T) read back as zeros, not as stale HBM
    hipMemset(R.dones, 0, T1 * B); hipMemset(R.firststeps, 0, T1 * B); hipMemset(R.rewards, 0, T1 * B * 4);
    for (int s = 0; s < c->S; ++s) CBM_HIP(hipEventCreateWithFlags(&R.ready[s], hipEventDisableTiming));
    CBM_HIP(hipEventCreateWithFlags(&R.consumed, hipEventDisableTiming));
  }
  // stream priorities (experiment knob, profiles/NOTES_r03_r04.md, round 4: nothing measurable either way): CBM_STREAM_PRIO=learner|actor|actor_hi|learner_lo
  int prio_lo = 0, prio_hi = 0;
  hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  const char* pe = getenv("CBM_STREAM_PRIO");
  int prio_actor = pe && !strcmp(pe, "actor") ? prio_hi : (pe && !strcmp(pe, "learner") ? prio_lo : 0);
  int prio_learner = pe && !strcmp(pe, "learner") ? prio_hi : (pe && !strcmp(pe, "actor") ? prio_lo : 0);
  if (pe && !strcmp(pe, "actor_hi")) { prio_actor = prio_hi; prio_learner = 0; }      // (actor / learner also push the other side to the LOW queue)
  if (pe && !strcmp(pe, "learner_lo")) { prio_actor = 0; prio_learner = prio_lo; }
  for (int s = 0; s < c->S; ++s) {
    Slot& sl = c->slots[s];
    CBM_HIP(hipStreamCreateWithPriority(&sl.stream, hipStreamNonBlocking, prio_actor));
    if (nature_ws_alloc(sl.ws, c->E, false, cfg->actor_dense_ksplit, cfg->network)) return -1;
    sl.ws.bf16_fwd = cfg->forward_bf16 != 0 && cfg->network == CBM_NET_NATURE;
    if (dalloc(&sl.env_state, (size_t)c->E) || dalloc(&sl.stats_dev, 2)) return -1;
    sl.pin_stride = ((size_t)c->E * 16 + 255) & ~(size_t)255;
    CBM_HIP(hipHostMalloc((void**)&sl.pin, sl.pin_stride * CBM_PIN_RING, hipHostMallocDefault));
    c->committed[s] = 0;
  }
  CBM_HIP(hipStreamCreateWithPriority(&c->lstream, hipStreamNonBlocking, prio_learner));
  const int lmax = c->MB > c->Bdev ? c->MB : c->Bdev;
  if (nature_ws_alloc(c->lws, lmax, true, cfg->actor_dense_ksplit, cfg->network)) return -1;
  c->lws.bf16_fwd = cfg->forward_bf16 != 0 && cfg->network == CBM_NET_NATURE;
  c->lws.bwd_split = cfg->backward_split;
  {   // conv1 at learner sizes: exact uint8 x split-bf16 products unless the fp32 chain is asked for (bit 0: forward, bit 1: weight gradient).
      // CBM_CONV1_EXACT=0|1 overrides the config for A/B runs (0 = both on the chain, 1 = both exact)
    const char* e = getenv("CBM_CONV1_EXACT");
    const int chain = e ? (e[0] == '0' ? 3 : 0) : cfg->conv1_fp32_chain;
    c->lws.conv1_exact_fwd = cfg->network == CBM_NET_NATURE && !(chain & 1);
    c->lws.conv1_exact_wgrad = cfg->network == CBM_NET_NATURE && !(chain & 2);
  }
  CBM_HIP(hipEventCreateWithFlags(&c->tail_ev, hipEventDisableTiming));
  CBM_HIP(hipEventCreateWithFlags(&c->ext_ev, hipEventDisableTiming));
  CBM_HIP(hipEventCreateWithFlags(&c->bwd_ev, hipEventDisableTiming));
  c->lws.tail_ev = c->tail_ev;
  CBM_HIP(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
  CBM_HIP(hipStreamCreateWithFlags(&c->iostream, hipStreamNonBlocking));
  { const char* ov = getenv("CBM_ALLREDUCE_OVERLAP"); c->comm_overlap = !(ov && !strcmp(ov, "0")); }
  if (c->accum > 1) { if (dalloc(&c->gacc, P)) return -1; hipMemset(c->gacc, 0, P * 4); }
  if (c->asyncB && dalloc(&c->advn, T1 * B)) return -1;
  if (dalloc(&c->adv, T1 * B) || dalloc(&c->target, T1 * B) || dalloc(&c->next_value, B) ||
      dalloc(&c->loss_partials, (size_t)4 * (lmax / 8 + 2) + 3 * B) || dalloc(&c->norm_partials, CBM_NORM_PARTS) ||
      false) return -1;
  {
    // permutation buffers: [epochs][n] rows and the batched scratch (~512 B per sample and (epoch, round) job) only where the batched path can run
    // — a PPO context with >= 2 epochs whose jobs fit CBM_PERM_BATCH_MAX; IMPALA never permutes, one epoch / too many jobs permute epoch by epoch
    const int n_perm = (int)(c->T * c->Bdev);
    c->perm_batched = is_ppo(c) && permutation_batch_ok(n_perm, c->epochs);
    const size_t rows = c->perm_batched ? (size_t)c->epochs : 1;
    const size_t scratch = c->perm_batched ? std::max(permutation_scratch_u64((int)(T1 * B)), permutation_batch_scratch_u64(n_perm, c->epochs))
                                           : permutation_scratch_u64((int)(T1 * B));
    if (dalloc(&c->perm, rows * T1 * B) || dalloc(&c->perm_tmp, rows * T1 * B) || dalloc(&c->ckeys, scratch)) return -1;
  }
  c->perm_cur = c->perm;
  if (!is_ppo(c)) {  // static minibatch index table: contiguous env-column chunks, all T+1 rows (impala:623-634)
    const int Bm = c->Bdev / c->nmicro;
    std::vector<int32_t> h((size_t)c->nmicro * c->MB);
    for (int mb = 0; mb < c->nmicro; ++mb)
      for (int t = 0; t < c->T1; ++t)
        for (int j = 0; j < Bm; ++j) h[(size_t)mb * c->MB + t * Bm + j] = t * c->Bdev + mb * Bm + j;
    if (dalloc(&c->impala_idx, h.size())) return -1;
    CBM_HIP(hipMemcpy(c->impala_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CBM_HIP(hipDeviceSynchronize());
  *out = c;
  return 0;
}

extern "C" int cbm_ctx_destroy(cbm_ctx* c);
extern "C" int cbm_ctx_create(const cbm_config* cfg, cbm_ctx** out) {
  cbm_ctx* partial = nullptr;
  const int rc = ctx_create_impl(cfg, out, &partial);
  if (rc != 0 && partial) {   // no leaks on a failed create (device memory of a half-built 9 GB ResNet workspace matters)
    const std::string msg = cbm_last_error();
    cbm_ctx_destroy(partial);
    cbm_set_error("%s", msg.c_str());
  }
  return rc;
}
extern "C" int cbm_ctx_destroy(cbm_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->cfg.device);
  hipDeviceSynchronize();
  cbm_ipc_close_all_impl(c);   // peer mappings first (a host that shares windows calls cbm_ipc_close_all on every peer, then a barrier, then this)
  void* ps[] = {c->params, c->opt_m, c->opt_v, c->adv, c->target, c->next_value, c->loss_partials, c->norm_partials,
                c->perm, c->perm_tmp, c->ckeys, c->impala_idx, c->gacc, c->advn};
  for (void* p : ps) if (p) hipFree(p);
  for (int i = 0; i < NPV; ++i) if (c->params_ready[i]) hipEventDestroy(c->params_ready[i]);
  for (int r = 0; r < c->cfg.ring_depth; ++r) {
    RingEntry& R = c->ring[r];
    for (int s = 0; s < c->S; ++s) if (R.ready[s]) hipEventDestroy(R.ready[s]);
    if (R.consumed) hipEventDestroy(R.consumed);
  }
  for (int s = 0; s < c->S; ++s) {
    nature_ws_free(c->slots[s].ws);
    if (c->slots[s].env_state) hipFree(c->slots[s].env_state);
    if (c->slots[s].stats_dev) hipFree(c->slots[s].stats_dev);
    if (c->slots[s].pin) hipHostFree(c->slots[s].pin);
    hipStreamDestroy(c->slots[s].stream);
  }
  nature_ws_free(c->lws);
  if (c->tail_ev) hipEventDestroy(c->tail_ev);
  if (c->ext_ev) hipEventDestroy(c->ext_ev);
  if (c->bwd_ev) hipEventDestroy(c->bwd_ev);
  cbm_comm_destroy_all(c);
  if (c->comm_prof_created) for (int i = 0; i < 4 * CBM_COMM_PROF_MAX; ++i) hipEventDestroy(c->comm_prof_ev[i]);
  for (int w = 0; w < CBM_WINDOWS; ++w) if (c->win[w]) hipFree(c->win[w]);   // the ring, the actor parameter versions, gradient / statistics / scratch
  if (c->cstream) hipStreamDestroy(c->cstream);
  if (c->iostream) hipStreamDestroy(c->iostream);
  hipStreamDestroy(c->lstream);
  delete c;
  return 0;
}

// ------------------------------------------------------------------------------------------ params / buffers
extern "C" int cbm_params_set(cbm_ctx* c, const float* h, int64_t n) {
  if (n != c->P) { cbm_set_error("param count %lld != %lld", (long long)n, (long long)c->P); return -1; }
  CBM_HIP(hipSetDevice(c->cfg.device));
  CBM_HIP(hipDeviceSynchronize());
  CBM_HIP(hipMemcpy(c->params, h, (size_t)n * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < NPV; ++i) CBM_HIP(hipMemcpy(c->actor_params[i], h, (size_t)n * 4, hipMemcpyHostToDevice));
  CBM_HIP(hipMemset(c->opt_m, 0, (size_t)n * 4));
  CBM_HIP(hipMemset(c->opt_v, 0, (size_t)n * 4));
  return 0;
}
extern "C" int cbm_params_get(cbm_ctx* c, float* h, int64_t n) {
  if (n != c->P) { cbm_set_error("param count mismatch"); return -1; }
  CBM_HIP(hipStreamSynchronize(c->lstream));
  CBM_HIP(hipMemcpy(h, c->params, (size_t)n * 4, hipMemcpyDeviceToHost));
  return cbm_comm_check_native(c);
}
extern "C" int cbm_actor_params_get(cbm_ctx* c, float* h, int64_t n) {
  if (n != c->P) { cbm_set_error("param count mismatch"); return -1; }
  CBM_HIP(hipDeviceSynchronize());
  CBM_HIP(hipMemcpy(h, c->actor_params[c->slots[0].pver % NPV], (size_t)n * 4, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int cbm_buffer(cbm_ctx* c, const char* name, int32_t ri, void** p, int64_t* nbytes) {
  const size_t TB = (size_t)c->T1 * c->Bdev;
  std::string n(name);
  if (ri < 0 || ri >= c->cfg.ring_depth) ri = 0;
  RingEntry& R = c->ring[ri];
  struct { const char* k; void* ptr; size_t sz; } tab[] = {
      {"params", c->params, (size_t)c->P * 4}, {"grads", c->grads, (size_t)c->P * 4}, {"opt_m", c->opt_m, (size_t)c->P * 4},
      {"opt_v", c->opt_v, (size_t)c->P * 4}, {"actor_params", c->actor_params[c->slots[0].pver % NPV], (size_t)c->P * 4},
      {"actor_params_latest", c->actor_params[c->updates_done % NPV], (size_t)c->P * 4},
      {"actor_params_v0", c->actor_params[0], (size_t)c->P * 4}, {"actor_params_v1", c->actor_params[1], (size_t)c->P * 4},
      {"actor_params_v2", c->actor_params[2], (size_t)c->P * 4},
      {"adv", c->adv, TB * 4}, {"target", c->target, TB * 4}, {"perm", c->perm, TB * 4}, {"next_value", c->next_value, (size_t)c->Bdev * 4},
      {"stats", c->stats_dev, (size_t)c->stat_rows * 8 * 4}, {"obs", R.obs, TB * CBM_FRAME}, {"actions", R.actions, TB * 4},
      {"logprobs", R.logprobs, TB * 4}, {"values", R.values, TB * 4}, {"rewards", R.rewards, TB * 4}, {"logits", R.logits, TB * c->A * 4},
      {"dones", R.dones, TB}, {"firststeps", R.firststeps, TB}, {"env_ids", R.env_ids, TB * 4}, {"adv_norm", c->advn, c->advn ? TB * 4 : 0}, {"lws_logits", c->lws.logits, (size_t)c->lws.maxB * 32 * 4},
      {"lws_value", c->lws.value, (size_t)c->lws.maxB * 4},
      // learner-workspace activations and their ReLU bit masks (tests / debugging)
      {"lws_act1", c->lws.act1, (size_t)c->lws.maxB * 12800 * 4}, {"lws_act2", c->lws.act2, (size_t)c->lws.maxB * 5184 * 4},
      {"lws_act3", c->lws.act3, (size_t)c->lws.maxB * 3136 * 4}, {"lws_mask1", c->lws.mask1, c->lws.mask1 ? (size_t)c->lws.maxB * 400 * 4 : 0},
      {"lws_mask2", c->lws.mask2, c->lws.mask2 ? (size_t)c->lws.maxB * 162 * 4 : 0}, {"lws_mask3", c->lws.mask3, c->lws.mask3 ? (size_t)c->lws.maxB * 98 * 4 : 0}};
  for (auto& e : tab)
    if (n == e.k) { *p = e.ptr; if (nbytes) *nbytes = (int64_t)e.sz; return 0; }
  cbm_set_error("unknown buffer '%s'", name);
  return -1;
}
extern "C" int cbm_copy_to_host(cbm_ctx* c, void* dst, const void* src, int64_t n) {
  CBM_HIP(hipDeviceSynchronize());
  CBM_HIP(hipMemcpy(dst, src, (size_t)n, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int cbm_copy_to_device(cbm_ctx* c, void* dst, const void* src, int64_t n) {
  CBM_HIP(hipDeviceSynchronize());
  CBM_HIP(hipMemcpy(dst, src, (size_t)n, hipMemcpyHostToDevice));
  return 0;
}
extern "C" int cbm_dev_alloc(int64_t nbytes, void** p) { CBM_HIP(hipMalloc(p, (size_t)nbytes)); return 0; }
extern "C" int cbm_dev_free(void* p) { CBM_HIP(hipFree(p)); return 0; }
extern "C" void* cbm_learner_stream(cbm_ctx* c) { return (void*)c->lstream; }
extern "C" int cbm_sync(cbm_ctx* c) { CBM_HIP(hipSetDevice(c->cfg.device)); CBM_HIP(hipDeviceSynchronize()); return cbm_comm_check_native(c); }

// ------------------------------------------------------------------------------------------ actor
extern "C" int cbm_actor_set_key(cbm_ctx* c, int32_t s, const uint32_t key[2]) { c->slots[s].key[0] = key[0]; c->slots[s].key[1] = key[1]; return 0; }
extern "C" int cbm_actor_get_key(cbm_ctx* c, int32_t s, uint32_t key[2]) { key[0] = c->slots[s].key[0]; key[1] = c->slots[s].key[1]; return 0; }

extern "C" int cbm_actor_env_reset_device(cbm_ctx* c, int32_t s, uint32_t seed) { return cbm_actor_env_reset_device_games(c, s, seed, 0); }
extern "C" int cbm_actor_env_reset_device_games(cbm_ctx* c, int32_t s, uint32_t seed, int32_t atari57_mix) {
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  RingEntry& R = c->ring[0];
  sl.env_seed = seed;
  launch_env_reset(seed, c->E, atari57_mix, sl.env_state, R.obs + (size_t)s * c->E * CBM_FRAME, CBM_FRAME, R.dones + s * c->E, R.firststeps + s * c->E, sl.stream);
  sl.env_inited = true;
  return 0;
}

static size_t row_off(const cbm_ctx* c, int t, int s) { return (size_t)t * c->Bdev + (size_t)s * c->E; }

extern "C" int cbm_actor_begin_rollout(cbm_ctx* c, int32_t s, int32_t concurrency, int32_t* policy_version) {
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  const int u = ++sl.rollout;
  const int need = concurrency ? (u >= 2 ? u - 2 : 0) : u - 1;
  const int depth = c->cfg.ring_depth;
  const int need_free = u - depth;  // ring entry reused: its previous rollout must be consumed
  if (!cbm_wait(c, [&] { const int d = c->updates_done.load(std::memory_order_acquire); return d >= need && d >= need_free; })) {
    cbm_set_error("context aborted"); return -4;
  }
  const int ri = (u - 1) % depth;
  if (need > 0) CBM_HIP(hipStreamWaitEvent(sl.stream, c->params_ready[need % NPV], 0));
  if (need_free > 0) CBM_HIP(hipStreamWaitEvent(sl.stream, c->ring[ri].consumed, 0));
  sl.pver = need;
  sl.ring = ri;
  sl.t = 0;
  if (u >= 2 && !c->asyncB) {   // (async rollouts start from whatever recv() returns next: nothing is carried, naturecnn:306-311)
    // carry the last row of the previous rollout to the head of this one: PPO next_obs/next_done (ppo:308-310),
    // IMPALA the whole bootstrap transition (impala:416)
    RingEntry& Pv = c->ring[(u - 2) % depth];
    RingEntry& R = c->ring[ri];
    const size_t so = row_off(c, c->T, s), d0 = row_off(c, 0, s);
    const size_t E = (size_t)c->E;
    CBM_HIP(hipMemcpyAsync(R.obs + d0 * CBM_FRAME, Pv.obs + so * CBM_FRAME, E * CBM_FRAME, hipMemcpyDeviceToDevice, sl.stream));
    CBM_HIP(hipMemcpyAsync(R.dones + d0, Pv.dones + so, E, hipMemcpyDeviceToDevice, sl.stream));
    CBM_HIP(hipMemcpyAsync(R.firststeps + d0, Pv.firststeps + so, E, hipMemcpyDeviceToDevice, sl.stream));
    if (!is_ppo(c)) {
      CBM_HIP(hipMemcpyAsync(R.actions + d0, Pv.actions + so, E * 4, hipMemcpyDeviceToDevice, sl.stream));
      CBM_HIP(hipMemcpyAsync(R.rewards + d0, Pv.rewards + so, E * 4, hipMemcpyDeviceToDevice, sl.stream));
      CBM_HIP(hipMemcpyAsync(R.logits + d0 * c->A, Pv.logits + so * c->A, E * c->A * 4, hipMemcpyDeviceToDevice, sl.stream));
      sl.t = 1;
    }
  }
  if (policy_version) *policy_version = need + 1;
  return 0;
}

// get_action_and_value on ring row t of slot s + Gumbel sampling, results stored into the row
// `env` (device env only): where the env's step with the sampled action reads and writes; when the forward pass ends in the per-frame tail
// launch that launch also steps the env (returns true), otherwise the caller launches env_step_kernel.
struct ActorEnvRows { size_t o_prev, o_next, o_reward; };
static bool actor_infer_row(cbm_ctx* c, int s, int t, const ActorEnvRows* env = nullptr) {
  Slot& sl = c->slots[s];
  RingEntry& R = c->ring[sl.ring];
  const size_t o = row_off(c, t, s);
  uint32_t n0, n1, s0, s1;  // key, subkey = jax.random.split(key)  (ppo:256)
  cbm_split_at(sl.key[0], sl.key[1], 2, 0, &n0, &n1);
  cbm_split_at(sl.key[0], sl.key[1], 2, 1, &s0, &s1);
  sl.key[0] = n0; sl.key[1] = n1;
  ActorSample smp = is_ppo(c) ? ActorSample{s0, s1, R.actions + o, R.logprobs + o, R.values + o, nullptr}
                              : ActorSample{s0, s1, R.actions + o, nullptr, nullptr, R.logits + o * c->A};
  if (env) {
    smp.env_seed = sl.env_seed; smp.env_max_steps = 27000;  // ATARI_MAX_FRAMES ppo:121-123
    smp.env_st = sl.env_state; smp.env_obs_prev = R.obs + env->o_prev * CBM_FRAME; smp.env_obs_next = R.obs + env->o_next * CBM_FRAME;
    smp.env_reward = R.rewards + env->o_reward; smp.env_done_next = R.dones + env->o_next; smp.env_firststep_next = R.firststeps + env->o_next;
  }
  const int done = nature_forward(c->L, c->actor_params[sl.pver % NPV], R.obs + o * CBM_FRAME, nullptr, c->E, c->cfg.actor_dense_ksplit, sl.ws,
                                  sl.stream, &smp);
  if (done) return done == 2;   // the forward pass ended in the fused reduce + heads + sampling (+ env step) launch
  if (is_ppo(c))
    launch_sample(sl.ws.logits, c->E, c->A, s0, s1, R.actions + o, R.logprobs + o, sl.ws.value, R.values + o, nullptr, sl.stream);
  else
    launch_sample(sl.ws.logits, c->E, c->A, s0, s1, R.actions + o, nullptr, nullptr, nullptr, R.logits + o * c->A, sl.stream);
  return false;
}

// small host array -> device through the slot's page-locked ring (see Slot::pin)
static int stage_small(Slot& sl, void* dst, const void* src, size_t nbytes) {
  uint8_t* p = sl.pin + (size_t)(sl.pin_cur++ % CBM_PIN_RING) * sl.pin_stride;
  memcpy(p, src, nbytes);
  CBM_HIP(hipMemcpyAsync(dst, p, nbytes, hipMemcpyHostToDevice, sl.stream));
  return 0;
}
extern "C" int cbm_actor_step_host(cbm_ctx* c, int32_t s, const uint8_t* obs, const uint8_t* done, const uint8_t* firststep,
                                   const float* reward_with_obs, int32_t* actions_out) {
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (sl.t > c->T) { cbm_set_error("rollout overrun: t=%d", sl.t); return -1; }
  RingEntry& R = c->ring[sl.ring];
  const size_t o = row_off(c, sl.t, s);
  const size_t E = (size_t)c->E;
  CBM_HIP(hipMemcpyAsync(R.obs + o * CBM_FRAME, obs, E * CBM_FRAME, hipMemcpyHostToDevice, sl.stream));
  if (done && stage_small(sl, R.dones + o, done, E)) return -1;
  if (firststep && stage_small(sl, R.firststeps + o, firststep, E)) return -1;
  if (reward_with_obs && stage_small(sl, R.rewards + o, reward_with_obs, E * 4)) return -1;
  actor_infer_row(c, s, sl.t);
  uint8_t* pa = sl.pin + (size_t)(sl.pin_cur++ % CBM_PIN_RING) * sl.pin_stride;   // actions come back through page-locked memory too
  CBM_HIP(hipMemcpyAsync(pa, R.actions + o, E * 4, hipMemcpyDeviceToHost, sl.stream));
  CBM_HIP(hipStreamSynchronize(sl.stream));  // the per-step D2H sync of ppo:317
  memcpy(actions_out, pa, E * 4);
  sl.t += 1;
  return cbm_launch_check();
}
// Async host-env step (naturecnn:346-367): the batch envpool.recv() returned — observations of `async_batch_size` envs, the reward /
// done that arrived WITH them and their env ids — goes into ring row t; get_action_and_value runs on it; actions come back for
// envs.send(action, env_id).
extern "C" int cbm_actor_step_async(cbm_ctx* c, int32_t s, const uint8_t* obs, const float* reward, const uint8_t* done, const int32_t* env_id,
                                    int32_t* actions_out) {
  if (!c->asyncB) { cbm_set_error("context was created without async_batch_size"); return -1; }
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (sl.t >= c->T) { cbm_set_error("rollout overrun: row %d of %d", sl.t, c->T); return -1; }
  for (int j = 0; j < c->E; ++j)
    if (env_id[j] < 0 || env_id[j] >= c->NE) { cbm_set_error("env_id %d outside [0,%d)", env_id[j], c->NE); return -1; }
  RingEntry& R = c->ring[sl.ring];
  const size_t o = row_off(c, sl.t, s), E = (size_t)c->E;
  CBM_HIP(hipMemcpyAsync(R.obs + o * CBM_FRAME, obs, E * CBM_FRAME, hipMemcpyHostToDevice, sl.stream));
  CBM_HIP(hipMemcpyAsync(R.dones + o, done, E, hipMemcpyHostToDevice, sl.stream));
  CBM_HIP(hipMemcpyAsync(R.rewards + o, reward, E * 4, hipMemcpyHostToDevice, sl.stream));
  CBM_HIP(hipMemcpyAsync(R.env_ids + o, env_id, E * 4, hipMemcpyHostToDevice, sl.stream));
  actor_infer_row(c, s, sl.t);
  CBM_HIP(hipMemcpyAsync(actions_out, R.actions + o, E * 4, hipMemcpyDeviceToHost, sl.stream));
  CBM_HIP(hipStreamSynchronize(sl.stream));
  sl.t += 1;
  return 0;
}
extern "C" int cbm_host_register(cbm_ctx* c, void* ptr, int64_t nbytes) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (!ptr || nbytes <= 0) { cbm_set_error("cbm_host_register: empty buffer"); return -1; }
  CBM_HIP(hipHostRegister(ptr, (size_t)nbytes, hipHostRegisterDefault));
  return 0;
}
extern "C" int cbm_host_unregister(cbm_ctx* c, void* ptr) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  CBM_HIP(hipHostUnregister(ptr));
  return 0;
}
extern "C" int cbm_actor_record_host(cbm_ctx* c, int32_t s, const float* reward) {
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  RingEntry& R = c->ring[sl.ring];
  const size_t o = row_off(c, sl.t - 1, s);
  return stage_small(sl, R.rewards + o, reward, (size_t)c->E * 4);
}

extern "C" int cbm_actor_rollout_device(cbm_ctx* c, int32_t s, int32_t nsteps) {
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (!sl.env_inited) { cbm_set_error("device env not reset: call cbm_actor_env_reset_device first"); return -1; }
  if (c->asyncB) { cbm_set_error("async_batch_size runs with a host env (cbm_actor_step_async); the device env is synchronous"); return -1; }
  RingEntry& R = c->ring[sl.ring];
  const int max_steps = 27000;  // ATARI_MAX_FRAMES ppo:121-123
  if (is_ppo(c)) {
    for (int i = 0; i < nsteps; ++i) {
      const int t = sl.t;
      if (t >= c->T) { cbm_set_error("rollout overrun"); return -1; }
      const size_t o = row_off(c, t, s), o1 = row_off(c, t + 1, s);
      const ActorEnvRows er{o, o1, o};
      if (!actor_infer_row(c, s, t, &er))
        launch_env_step(sl.env_seed, c->E, max_steps, R.actions + o, sl.env_state, R.obs + o * CBM_FRAME, R.obs + o1 * CBM_FRAME, R.rewards + o,
                        R.dones + o1, R.firststeps + o1, sl.stream);
      sl.t += 1;
    }
  } else {
    // IMPALA: row t holds what arrived with obs_t; stepping with action_t fills row t+1 (impala:352-384).
    if (sl.t == 1 && sl.rollout >= 2) {  // carried transition: its action has not been sent to the env yet
      const size_t o = row_off(c, 0, s), o1 = row_off(c, 1, s);
      launch_env_step(sl.env_seed, c->E, max_steps, R.actions + o, sl.env_state, R.obs + o * CBM_FRAME, R.obs + o1 * CBM_FRAME, R.rewards + o1,
                      R.dones + o1, R.firststeps + o1, sl.stream);
    }
    for (int i = 0; i < nsteps; ++i) {
      const int t = sl.t;
      if (t > c->T) { cbm_set_error("rollout overrun"); return -1; }
      if (t < c->T) {
        const size_t o = row_off(c, t, s), o1 = row_off(c, t + 1, s);
        const ActorEnvRows er{o, o1, o1};
        if (!actor_infer_row(c, s, t, &er))
          launch_env_step(sl.env_seed, c->E, max_steps, R.actions + o, sl.env_state, R.obs + o * CBM_FRAME, R.obs + o1 * CBM_FRAME, R.rewards + o1,
                          R.dones + o1, R.firststeps + o1, sl.stream);
      } else {
        actor_infer_row(c, s, t);
      }
      sl.t += 1;
    }
  }
  return cbm_launch_check();
}

extern "C" int cbm_actor_commit(cbm_ctx* c, int32_t s, const uint8_t* next_obs, const uint8_t* next_done) {
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  RingEntry& R = c->ring[sl.ring];
  if (next_obs) {
    const size_t o = row_off(c, c->T, s);
    CBM_HIP(hipMemcpyAsync(R.obs + o * CBM_FRAME, next_obs, (size_t)c->E * CBM_FRAME, hipMemcpyHostToDevice, sl.stream));
    if (next_done) CBM_HIP(hipMemcpyAsync(R.dones + o, next_done, (size_t)c->E, hipMemcpyHostToDevice, sl.stream));
    CBM_HIP(hipStreamSynchronize(sl.stream));  // host buffers may be reused by the caller
  }
  CBM_HIP(hipEventRecord(R.ready[s], sl.stream));
  cbm_publish(c, c->committed[s], sl.rollout);
  return 0;
}

extern "C" int cbm_actor_episode_stats(cbm_ctx* c, int32_t s, float* avg_return, float* avg_length) {
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  float h[2] = {0, 0};
  launch_env_stats(sl.env_state, c->E, sl.stats_dev, sl.stream);
  CBM_HIP(hipMemcpyAsync(h, sl.stats_dev, 8, hipMemcpyDeviceToHost, sl.stream));
  CBM_HIP(hipStreamSynchronize(sl.stream));
  if (avg_return) *avg_return = h[0];
  if (avg_length) *avg_length = h[1];
  return 0;
}

// ------------------------------------------------------------------------------------------ split topologies
// a0-l1,2,3 style runs (ppo:97-100, README.md:62): actor and learners are different processes / GPUs.  A learner-only
// ctx treats its "slots" as ingest ports: the host copies (or RCCL-receives) a rollout shard straight into the ring
// entry returned by cbm_ingest_begin and publishes it with cbm_ingest_commit; an actor-only ctx receives parameter
// versions through cbm_params_publish_external (the params_queue.put of ppo:721-725 coming from another process).
extern "C" int cbm_ingest_begin(cbm_ctx* c, int32_t s, int32_t* ring_index) {
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  const int u = ++sl.rollout;
  const int depth = c->cfg.ring_depth;
  if (!cbm_wait(c, [&] { return c->updates_done.load(std::memory_order_acquire) >= u - depth; })) { cbm_set_error("context aborted"); return -4; }
  sl.ring = (u - 1) % depth;
  if (u > depth) CBM_HIP(hipStreamWaitEvent(sl.stream, c->ring[sl.ring].consumed, 0));
  CBM_HIP(hipStreamSynchronize(sl.stream));  // the entry is free on the device too: the caller may overwrite it from any stream
  if (ring_index) *ring_index = sl.ring;
  return 0;
}
extern "C" int cbm_ingest_commit(cbm_ctx* c, int32_t s) {
  Slot& sl = c->slots[s];
  CBM_HIP(hipSetDevice(c->cfg.device));
  // contract: the caller's shard copies into the entry have COMPLETED (it synchronised the stream it issued them on)
  CBM_HIP(hipEventRecord(c->ring[sl.ring].ready[s], sl.stream));
  cbm_publish(c, c->committed[s], sl.rollout);
  return 0;
}
extern "C" int cbm_params_publish_external(cbm_ctx* c, const float* dev_params, int64_t n) {
  if (n != c->P) { cbm_set_error("param count mismatch"); return -1; }
  CBM_HIP(hipSetDevice(c->cfg.device));
  const int v = c->updates_done + 1;
  CBM_HIP(hipMemcpyAsync(c->actor_params[v % NPV], dev_params, (size_t)n * 4, hipMemcpyDeviceToDevice, c->lstream));
  CBM_HIP(hipEventRecord(c->params_ready[v % NPV], c->lstream));
  CBM_HIP(hipStreamSynchronize(c->lstream));  // the source buffer may be reused by the caller
  cbm_publish(c, c->updates_done, v);
  return 0;
}
extern "C" int cbm_ctx_abort(cbm_ctx* c) {
  c->aborted.store(true, std::memory_order_release);
  c->epoch.fetch_add(1, std::memory_order_release);
  c->epoch.notify_all();
  return 0;
}
extern "C" void* cbm_actor_stream(cbm_ctx* c, int32_t s) { return (void*)c->slots[s].stream; }
extern "C" int cbm_actor_ring_index(cbm_ctx* c, int32_t s) { return c->slots[s].ring; }

// ------------------------------------------------------------------------------------------ learner
extern "C" int cbm_learner_wait(cbm_ctx* c) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  const int v = c->updates_done + 1;
  if (!cbm_wait(c, [&] { for (int s = 0; s < c->S; ++s) if (c->committed[s].load(std::memory_order_acquire) < v) return false; return true; })) {
    cbm_set_error("context aborted"); return -4;
  }
  RingEntry& R = c->ring[(v - 1) % c->cfg.ring_depth];
  for (int s = 0; s < c->S; ++s) CBM_HIP(hipStreamWaitEvent(c->lstream, R.ready[s], 0));
  return 0;
}

static RingEntry& cur_ring(cbm_ctx* c) { return c->ring[c->updates_done % c->cfg.ring_depth]; }

extern "C" int cbm_learner_prepare(cbm_ctx* c, uint32_t key[2]) {
  (void)key;
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (!is_ppo(c)) return 0;
  RingEntry& R = cur_ring(c);
  const int B = c->Bdev, T = c->T;
  if (c->asyncB) {   // no bootstrap observation, env-id-indexed returns, normalisation per minibatch later (naturecnn:254-256,540)
    launch_gae_async(R.env_ids, R.rewards, R.values, R.dones, T, B, c->NE, c->cfg.gamma, c->cfg.gae_lambda, c->adv, c->target, c->lstream);
    return 0;
  }
  // compute_gae ppo:543-560: bootstrap value with the LEARNER's params on next_obs (row T)
  CbmProf* const pf = c->lws.prof;
  uint32_t* const m1 = c->lws.mask1;
  c->lws.prof = nullptr;   // the per-kernel timers are for the minibatch-size launches, not for this 1-row bootstrap pass
  c->lws.mask1 = nullptr;  // no backward follows it either: without ReLU masks wanted a pass of <= 512 frames takes the actor's small-batch kernels (same bits, 110 -> 60 us)
  nature_forward(c->L, c->params, R.obs + (size_t)T * B * CBM_FRAME, nullptr, B, c->cfg.actor_dense_ksplit, c->lws, c->lstream);
  c->lws.prof = pf;
  c->lws.mask1 = m1;
  CBM_HIP(hipMemcpyAsync(c->next_value, c->lws.value, (size_t)B * 4, hipMemcpyDeviceToDevice, c->lstream));
  launch_gae(R.rewards, R.values, R.dones, c->next_value, R.dones + (size_t)T * B, T, B, c->cfg.gamma, c->cfg.gae_lambda, c->adv, c->target, c->lstream);
  if (c->cfg.norm_adv) launch_advnorm(c->adv, T, B, c->nmb, c->lstream);
  return 0;
}

static int learner_epoch_perm(cbm_ctx* c, uint32_t key[2]) {
  // key, subkey = split(key); perm = permutation(subkey, N)   (ppo:599,606)
  uint32_t n0, n1, s0, s1;
  cbm_split_at(key[0], key[1], 2, 0, &n0, &n1);
  cbm_split_at(key[0], key[1], 2, 1, &s0, &s1);
  key[0] = n0; key[1] = n1;
  const uint32_t sub[2] = {s0, s1};
  launch_permutation(sub, c->T * c->Bdev, c->perm, c->perm_tmp, c->ckeys, c->lstream);
  c->perm_cur = c->perm;
  return 0;
}
// The whole-update call knows every epoch's subkey before its first minibatch: all permutations in four launches (pointwise.hip) instead of six
// per epoch on the learner stream; epoch e then reads rows [e][...] of c->perm.  The key advances exactly as epoch-by-epoch calls advance it.
static bool learner_all_epoch_perms(cbm_ctx* c, uint32_t key[2]) {
  if (!c->perm_batched || c->epochs > CBM_PERM_BATCH_MAX) return false;
  uint32_t subs[CBM_PERM_BATCH_MAX][2];
  uint32_t k0 = key[0], k1 = key[1];
  for (int e = 0; e < c->epochs; ++e) {
    uint32_t n0, n1;
    cbm_split_at(k0, k1, 2, 1, &subs[e][0], &subs[e][1]);
    cbm_split_at(k0, k1, 2, 0, &n0, &n1);
    k0 = n0; k1 = n1;
  }
  if (!launch_permutations_batch(subs, c->epochs, c->T * c->Bdev, c->perm, c->perm_tmp, c->ckeys, c->lstream)) return false;
  key[0] = k0; key[1] = k1;
  return true;
}

// Small minibatches (<= 1024 frames, e.g. IMPALA's default 21 x 30) leave the 3136 -> 512 dense with ~80 blocks and a 98-chunk
// serial K loop (119 us at 630 frames, 17 TF); they use the actor's K split instead, which also makes the learner's logits the
// actor's bit for bit.  Large minibatches keep the single chain.
static int learner_ksplit(const cbm_ctx* c) { return c->MB <= 1024 ? c->cfg.actor_dense_ksplit : 1; }
extern "C" int cbm_learner_minibatch_grad(cbm_ctx* c, int32_t epoch, int32_t mb) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  c->lws.tail_ev = c->comms[CBM_COMM_LEARNERS].nranks ? c->tail_ev : nullptr;   // the gradient tail is finished early only when an all-reduce waits for it
  RingEntry& R = cur_ring(c);
  float* stats = c->stats_dev + (size_t)(epoch * c->nmicro + mb) * 8;
  if (is_ppo(c)) {
    const int32_t* idx = c->perm_cur + (size_t)mb * c->MB;
    c->lws.skip_heads = ppo_heads_fusable(c->L);
    nature_forward(c->L, c->params, R.obs, idx, c->MB, learner_ksplit(c), c->lws, c->lstream);
    const float* adv = c->adv;
    if (c->asyncB && c->cfg.norm_adv) { launch_mb_advnorm(c->adv, idx, c->MB, c->advn, c->lstream); adv = c->advn; }
    if (c->lws.skip_heads) {   // heads forward + loss + heads dgrad in one launch (set around this forward / backward pair only)
      launch_ppo_heads_fused(c->L, c->params, c->lws, c->MB, idx, R.actions, R.logprobs, adv, c->target, c->cfg.clip_coef, c->cfg.ent_coef, c->cfg.vf_coef,
                             c->loss_partials, stats, c->lstream);
    } else {
      launch_ppo_loss(c->lws.logits, c->lws.value, c->MB, c->A, idx, R.actions, R.logprobs, adv, c->target, c->cfg.clip_coef, c->cfg.ent_coef,
                      c->cfg.vf_coef, c->lws.dzv, c->loss_partials, stats, c->lstream);
    }
    nature_backward(c->L, c->params, R.obs, idx, c->MB, c->lws, c->grads, c->lstream);
    flush_pending_stats(c->lws, c->lstream);   // (only if the backward pass ended early)
    c->lws.skip_heads = false;
  } else {
    const int Bm = c->Bdev / c->nmicro;
    const int32_t* idx = c->impala_idx + (size_t)mb * c->MB;
    nature_forward(c->L, c->params, R.obs, idx, c->MB, learner_ksplit(c), c->lws, c->lstream);
    launch_impala_loss(c->lws.logits, c->lws.value, R.logits, R.actions, R.rewards, R.dones, R.firststeps, c->T1, Bm, c->A, mb * Bm, c->Bdev,
                       c->cfg.gamma, c->cfg.vf_coef, c->cfg.ent_coef, c->lws.dzv, c->loss_partials, stats, c->lstream);
    // The bootstrap row (t = T: the minibatch's LAST Bm frames) enters the loss through its value only and receives no gradient (impala:577-590): its
    // dL/d(logits, value) rows are zeros, so the backward pass runs on the first T * Bm frames — 3840 of 3870 at E = 120, T = 128, which is also
    // what the frame-resident kernels want (15 frames on each of 256 CUs instead of 16 on 242).
    nature_backward(c->L, c->params, R.obs, idx, c->MB - Bm, c->lws, c->grads, c->lstream);
  }
  return cbm_launch_check();
}

// Overlapping pmean(grads) (ppo:628) with the backward pass.  The flat gradient is laid out conv1 | conv2 | conv3 | dense | actor | critic and
// the backward pass produces it from the back: once the dense weight gradient is reduced, the tail [grad_tail_offset, P) — 95 % of the
// bytes (Nature) / 91 % (ResNet) — is final while the conv dgrad / wgrad kernels are still to run.  A data-parallel host all-reduces that
// tail on its own communication stream as soon as the event fires, the small head after the backward pass, and joins the streams.
extern "C" int64_t cbm_learner_grad_tail_offset(cbm_ctx* c) { return c->L.w[3]; }
// optax.MultiSteps (0.1.4): acc <- (g - acc)/(mini_step+1) + acc; the k-th micro-batch hands the mean to the inner optimizer and clears acc
extern "C" int cbm_learner_accumulate(cbm_ctx* c, int32_t mini_step, float grad_div) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (c->accum <= 1) { cbm_set_error("context was created without gradient accumulation"); return -1; }
  launch_grad_accumulate(c->grads, c->gacc, c->P, mini_step, mini_step == c->accum - 1, grad_div, c->lstream);
  return 0;
}
extern "C" int cbm_learner_optimizer_step(cbm_ctx* c, float lr, float bc1, float bc2, float grad_div) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (is_ppo(c))
    launch_adam(c->params, c->grads, c->opt_m, c->opt_v, c->P, c->cfg.max_grad_norm, lr, c->cfg.adam_b1, c->cfg.adam_b2, c->cfg.adam_eps, bc1, bc2,
                grad_div, c->norm_partials, c->lstream);
  else
    launch_rmsprop(c->params, c->grads, c->opt_m, c->P, c->cfg.max_grad_norm, lr, c->cfg.rms_decay, c->cfg.rms_eps, grad_div, c->norm_partials,
                   c->lstream);
  return 0;
}

extern "C" int cbm_learner_finish(cbm_ctx* c, float* stats_out) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  const int v = c->updates_done + 1;
  RingEntry& R = cur_ring(c);
  // publish params version v to the actors (ppo:721-725): D2D copy + event, no host sync
  CBM_HIP(hipMemcpyAsync(c->actor_params[v % NPV], c->params, (size_t)c->P * 4, hipMemcpyDeviceToDevice, c->lstream));
  CBM_HIP(hipEventRecord(c->params_ready[v % NPV], c->lstream));
  CBM_HIP(hipEventRecord(R.consumed, c->lstream));
  if (stats_out) {
    const int w = is_ppo(c) ? 5 : 4;
    if (cbm_learner_allreduce_stats_impl(c)) return -1;   // pmean over the learners (ppo:649-653); no-op without a communicator
    const float inv = c->comms[CBM_COMM_LEARNERS].nranks ? 1.0f / (float)c->comms[CBM_COMM_LEARNERS].nranks : 1.0f;
    std::vector<float> h((size_t)c->stat_rows * 8);
    CBM_HIP(hipMemcpyAsync(h.data(), c->stats_dev, h.size() * 4, hipMemcpyDeviceToHost, c->lstream));
    CBM_HIP(hipStreamSynchronize(c->lstream));
    if (cbm_comm_check_native(c)) return -1;
    for (int r = 0; r < c->stat_rows; ++r) for (int q = 0; q < w; ++q) stats_out[r * w + q] = h[(size_t)r * 8 + q] * inv;
  }
  else if (c->comms[CBM_COMM_LEARNERS].native) {
    // a timed-out flag wait (a dead peer) leaves an unreduced gradient behind: with the native backend the update's outcome is checked before
    // version v is announced, statistics or not (RCCL reports its failures through its own calls)
    CBM_HIP(hipStreamSynchronize(c->lstream));
    if (cbm_comm_check_native(c)) return -1;
  }
  cbm_publish(c, c->updates_done, v);
  return 0;
}

extern "C" int cbm_learner_update(cbm_ctx* c, uint32_t key[2], const float* lrs, const float* bc1, const float* bc2, int32_t n_opt_steps,
                                  float* stats_out) {
  if (n_opt_steps != c->epochs * c->nmb) { cbm_set_error("n_opt_steps must be epochs*minibatches = %d", c->epochs * c->nmb); return -1; }
  if (cbm_learner_prepare(c, key)) return -1;
  int step = 0;
  const bool all_perms = is_ppo(c) && learner_all_epoch_perms(c, key);
  for (int e = 0; e < c->epochs; ++e) {
    if (all_perms) c->perm_cur = c->perm + (size_t)e * c->T * c->Bdev;
    else if (is_ppo(c)) learner_epoch_perm(c, key);
    for (int mb = 0; mb < c->nmicro; ++mb) {
      if (cbm_learner_minibatch_grad(c, e, mb)) return -1;
      float grad_div = 1.0f;
      if (cbm_learner_allreduce_grads_impl(c, &grad_div)) return -1;   // pmean over the learner GPUs (ppo:628); nothing on one GPU
      if (c->accum > 1) {
        if (cbm_learner_accumulate(c, mb % c->accum, grad_div)) return -1;
        if (mb % c->accum != c->accum - 1) continue;
        grad_div = 1.0f;
      }
      if (cbm_learner_optimizer_step(c, lrs[step], bc1 ? bc1[step] : 1.0f, bc2 ? bc2[step] : 1.0f, grad_div)) return -1;
      ++step;
    }
  }
  return cbm_learner_finish(c, stats_out);
}

// epoch permutation for the split (data-parallel) form
extern "C" int cbm_learner_epoch_begin(cbm_ctx* c, uint32_t key[2]) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (is_ppo(c)) return learner_epoch_perm(c, key);
  return 0;
}

// ------------------------------------------------------------------------------------------ profiling
extern "C" int cbm_profile_select(cbm_ctx* c, int32_t kernel_id) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (!c->prof.created) {
    for (int i = 0; i < 2 * CBM_PROF_MAX; ++i) CBM_HIP(hipEventCreate(&c->prof.ev[i]));
    c->prof.created = true;
  }
  if (kernel_id == CBM_PROF_PAUSE) {   // stop recording, keep what was recorded for the next cbm_profile_read*
    c->lws.prof = nullptr;
    return 0;
  }
  c->prof.sel = kernel_id;
  c->prof.n = 0;
  c->lws.prof = (kernel_id >= 0 || kernel_id == CBM_PROF_ALL) ? &c->prof : nullptr;
  return 0;
}
extern "C" int cbm_profile_read(cbm_ctx* c, double* total_ms, int32_t* count) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  CBM_HIP(hipStreamSynchronize(c->lstream));
  double tot = 0.0;
  for (int i = 0; i < c->prof.n; ++i) {
    float ms = 0.0f;
    CBM_HIP(hipEventElapsedTime(&ms, c->prof.ev[2 * i], c->prof.ev[2 * i + 1]));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (count) *count = c->prof.n;
  c->prof.n = 0;
  return 0;
}

extern "C" int cbm_profile_read_all(cbm_ctx* c, double* total_ms, int32_t* count, int32_t n_ids) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  CBM_HIP(hipStreamSynchronize(c->lstream));
  for (int k = 0; k < n_ids; ++k) { total_ms[k] = 0.0; count[k] = 0; }
  for (int i = 0; i < c->prof.n; ++i) {
    float ms = 0.0f;
    CBM_HIP(hipEventElapsedTime(&ms, c->prof.ev[2 * i], c->prof.ev[2 * i + 1]));
    const int k = c->prof.kid[i];
    if (k >= 0 && k < n_ids) { total_ms[k] += ms; count[k] += 1; }
  }
  c->prof.n = 0;
  return 0;
}

extern "C" int cbm_profile_kernel_name(cbm_ctx* c, int32_t kernel_id, char* buf, int32_t buf_len) {
  if (kernel_id < 0 || kernel_id >= K_NUM || !buf || buf_len < 1) { cbm_set_error("cbm_profile_kernel_name: bad arguments"); return -1; }
  const char* k = c->prof.kernel[kernel_id];
  const char* f = c->prof.functor[kernel_id];
  if (k) snprintf(buf, (size_t)buf_len, "%s%s%s", k, f && f[0] ? " " : "", f ? f : "");
  if (!k) buf[0] = 0;
  return 0;
}

// ------------------------------------------------------------------------------------------ pure functions
static int check_B(cbm_ctx* c, int B) {
  if (B > c->lws.maxB) { cbm_set_error("B=%d exceeds the learner workspace (%d frames)", B, c->lws.maxB); return -1; }
  return 0;
}
extern "C" int cbm_forward(cbm_ctx* c, const float* params, const uint8_t* obs, const int32_t* idx, int32_t B, int32_t ksplit, float* logits,
                           float* value) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (check_B(c, B)) return -1;
  if (ksplit > 1 && (B > 1024 || ksplit != c->lws.dense_part_ksplit || !c->lws.dense_part)) {
    // a split-K plan needs its partial buffer; allocate on demand for tests
    if (c->lws.dense_part) hipFree(c->lws.dense_part);
    CBM_HIP(hipMalloc((void**)&c->lws.dense_part, (size_t)ksplit * c->lws.maxB * c->L.hid * 4));
    c->lws.dense_part_ksplit = ksplit;
  }
  nature_forward(c->L, params, obs, idx, B, ksplit, c->lws, c->lstream);
  if (logits) CBM_HIP(hipMemcpyAsync(logits, c->lws.logits, (size_t)B * c->A * 4, hipMemcpyDeviceToDevice, c->lstream));
  if (value) CBM_HIP(hipMemcpyAsync(value, c->lws.value, (size_t)B * 4, hipMemcpyDeviceToDevice, c->lstream));
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return cbm_launch_check();
}
extern "C" int cbm_sample(cbm_ctx* c, const float* logits, int32_t B, const uint32_t sub[2], int32_t* actions, float* logprobs) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  launch_sample(logits, B, c->A, sub[0], sub[1], actions, logprobs, nullptr, nullptr, nullptr, c->lstream);
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return 0;
}
extern "C" int cbm_gae(cbm_ctx* c, const float* rewards, const float* values, const uint8_t* dones, const float* next_value,
                       const uint8_t* next_done, int32_t T, int32_t B, float* adv, float* target) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  launch_gae(rewards, values, dones, next_value, next_done, T, B, c->cfg.gamma, c->cfg.gae_lambda, adv, target, c->lstream);
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return 0;
}
extern "C" int cbm_vtrace(cbm_ctx* c, const float* v_tm1, const float* v_t, const float* r_t, const float* disc_t, const float* rho_tm1, int32_t T,
                          int32_t B, float* errors, float* pg_adv, float* q_est) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  launch_vtrace(v_tm1, v_t, r_t, disc_t, rho_tm1, T, B, errors, pg_adv, q_est, c->lstream);
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return 0;
}
extern "C" int cbm_gae_async(cbm_ctx* c, const int32_t* env_ids, const float* rewards, const float* values, const uint8_t* dones, int32_t R,
                             int32_t B, int32_t num_envs, float* adv, float* target) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  launch_gae_async(env_ids, rewards, values, dones, R, B, num_envs, c->cfg.gamma, c->cfg.gae_lambda, adv, target, c->lstream);
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return 0;
}
extern "C" int cbm_mb_advnorm(cbm_ctx* c, const float* adv, const int32_t* idx, int32_t n, float* out) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  launch_mb_advnorm(adv, idx, n, out, c->lstream);
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return 0;
}
extern "C" int cbm_advnorm(cbm_ctx* c, float* adv, int32_t T, int32_t B, int32_t groups) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  launch_advnorm(adv, T, B, groups, c->lstream);
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return 0;
}
extern "C" int cbm_permutation(cbm_ctx* c, const uint32_t key[2], int32_t n, int32_t* perm) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  int32_t* tmp = nullptr; uint64_t* ck = nullptr;
  CBM_HIP(hipMalloc((void**)&tmp, (size_t)n * 4));
  CBM_HIP(hipMalloc((void**)&ck, permutation_scratch_u64(n) * 8));
  launch_permutation(key, n, perm, tmp, ck, c->lstream);
  CBM_HIP(hipStreamSynchronize(c->lstream));
  hipFree(tmp); hipFree(ck);
  return 0;
}
extern "C" int cbm_ppo_loss_grad(cbm_ctx* c, const float* params, const uint8_t* obs, const int32_t* idx, int32_t N, const int32_t* actions,
                                 const float* old_logprob, const float* adv, const float* target, float* stats5, float* grads, float* logits_out,
                                 float* value_out) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  if (check_B(c, N)) return -1;
  c->lws.skip_heads = ppo_heads_fusable(c->L);
  nature_forward(c->L, params, obs, idx, N, 1, c->lws, c->lstream);
  if (c->lws.skip_heads)
    launch_ppo_heads_fused(c->L, params, c->lws, N, nullptr, actions, old_logprob, adv, target, c->cfg.clip_coef, c->cfg.ent_coef, c->cfg.vf_coef,
                           c->loss_partials, stats5, c->lstream);
  else
    launch_ppo_loss(c->lws.logits, c->lws.value, N, c->A, nullptr, actions, old_logprob, adv, target, c->cfg.clip_coef, c->cfg.ent_coef,
                    c->cfg.vf_coef, c->lws.dzv, c->loss_partials, stats5, c->lstream);
  if (grads) nature_backward(c->L, params, obs, idx, N, c->lws, grads, c->lstream);
  flush_pending_stats(c->lws, c->lstream);   // no backward pass ran (or it ended early): the statistics take their own launch
  c->lws.skip_heads = false;
  if (logits_out) CBM_HIP(hipMemcpyAsync(logits_out, c->lws.logits, (size_t)N * c->A * 4, hipMemcpyDeviceToDevice, c->lstream));
  if (value_out) CBM_HIP(hipMemcpyAsync(value_out, c->lws.value, (size_t)N * 4, hipMemcpyDeviceToDevice, c->lstream));
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return cbm_launch_check();
}
extern "C" int cbm_impala_loss_grad(cbm_ctx* c, const float* params, const uint8_t* obs, const int32_t* idx, int32_t T1, int32_t Bm,
                                    const float* mu_logits, const int32_t* actions, const float* rewards, const uint8_t* dones,
                                    const uint8_t* firststeps, float* stats4, float* grads) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  const int N = T1 * Bm;
  if (check_B(c, N)) return -1;
  nature_forward(c->L, params, obs, idx, N, 1, c->lws, c->lstream);
  float* partials = nullptr;
  CBM_HIP(hipMalloc((void**)&partials, (size_t)Bm * 3 * 4 + 64));
  launch_impala_loss(c->lws.logits, c->lws.value, mu_logits, actions, rewards, dones, firststeps, T1, Bm, c->A, 0, Bm, c->cfg.gamma, c->cfg.vf_coef,
                     c->cfg.ent_coef, c->lws.dzv, partials, stats4, c->lstream);
  if (grads) nature_backward(c->L, params, obs, idx, N - Bm, c->lws, grads, c->lstream);   // (the bootstrap row carries no gradient: see cbm_learner_minibatch_grad)
  CBM_HIP(hipStreamSynchronize(c->lstream));
  hipFree(partials);
  return cbm_launch_check();
}
extern "C" int cbm_adam_step(cbm_ctx* c, float* p, const float* g, float* m, float* v, int64_t n, float max_norm, float lr, float bc1, float bc2,
                             float grad_div) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  launch_adam(p, g, m, v, n, max_norm, lr, c->cfg.adam_b1, c->cfg.adam_b2, c->cfg.adam_eps, bc1, bc2, grad_div, c->norm_partials, c->lstream);
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return 0;
}
extern "C" int cbm_rmsprop_step(cbm_ctx* c, float* p, const float* g, float* nu, int64_t n, float max_norm, float lr, float grad_div) {
  CBM_HIP(hipSetDevice(c->cfg.device));
  launch_rmsprop(p, g, nu, n, max_norm, lr, c->cfg.rms_decay, c->cfg.rms_eps, grad_div, c->norm_partials, c->lstream);
  CBM_HIP(hipStreamSynchronize(c->lstream));
  return 0;
}
