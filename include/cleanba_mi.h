/* cleanba_mi.h — C ABI of libcleanba_mi.so, the MI355X (gfx950) implementation of cleanba's
 * rollout + update hot path.  extern "C", plain pointers and sizes, no torch types.
 *
 * The reference (vwxyzjn/cleanba, /root/reference) has no FFI: its seam is Python-level
 * (SURVEY.md §8b).  Each entry point below names the reference code it replaces
 * (file:line under /root/reference/cleanba/; "ppo" = cleanba_ppo.py, "impala" =
 * cleanba_impala.py, "naturecnn" = legacy_scripts/cleanba_ppo_envpool_impala_atari_wrapper_naturecnn.py).
 * INTEGRATION.md shows the ctypes stubs a cleanba maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; cbm_last_error() gives the message
 *    (thread-local).  Nothing throws, nothing calls back into Python.
 *  - the library owns ctx, HBM rollout ring, workspaces, streams, events.  Host pointers passed
 *    in are only read/written during the call.  Device pointers handed out by cbm_buffer() stay
 *    valid until cbm_ctx_destroy().
 *  - threading: one host thread per actor slot may call cbm_actor_* concurrently (each slot has
 *    its own HIP stream and ring producer index); one learner thread calls cbm_learner_*.
 *  - numerics: fp32 throughout, forward dot products are k-ascending fmaf chains
 *    (v_mfma_f32_32x32x2_f32); scalar math from cbm_math.h.
 */
#ifndef CLEANBA_MI_H
#define CLEANBA_MI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBM_ABI_VERSION 2

typedef struct cbm_ctx cbm_ctx;

enum { CBM_NET_NATURE = 0, CBM_NET_IMPALA_RESNET = 1 };
enum { CBM_ALGO_PPO = 0, CBM_ALGO_IMPALA = 1 };

/* Mirrors the fields of the reference `Args` that the path reads (ppo:34-118, impala:34-110). */
typedef struct {
  int32_t abi_version;       /* = CBM_ABI_VERSION */
  int32_t device;            /* HIP device ordinal (actor and learner share it: a0-l0) */
  int32_t network;           /* CBM_NET_*            (naturecnn:143-178 / ppo:149-189) */
  int32_t algo;              /* CBM_ALGO_*                                              */
  int32_t num_actions;       /* envs.single_action_space.n  (ppo:253)                   */
  int32_t local_num_envs;    /* --local-num-envs  E          (ppo:63)                   */
  int32_t num_actor_slots;   /* len(actor_device_ids)*num_actor_threads on this GPU (ppo:668-686) */
  int32_t num_steps;         /* --num-steps T                (ppo:67 / impala:67)       */
  int32_t num_minibatches;   /* ppo:77                                                  */
  int32_t update_epochs;     /* ppo:81 (IMPALA: 1)                                      */
  int32_t norm_adv;          /* ppo:83                                                  */
  int32_t ring_depth;        /* rollouts in flight per slot (>=2); replaces Queue(maxsize=1) ppo:672-673 */
  float gamma, gae_lambda;   /* ppo:72-75                                               */
  float clip_coef, ent_coef, vf_coef, max_grad_norm; /* ppo:85-92 / impala:81-86        */
  float adam_b1, adam_b2, adam_eps;                  /* optax.adam defaults, eps=1e-5 ppo:497 */
  float rms_decay, rms_eps;                          /* impala:534                       */
  int32_t actor_dense_ksplit;/* K segments of the 3136->512 dense when M<=128 (numerics spec) */
  int32_t forward_bf16;       /* build-only extension (BASELINE configs[2]): forward GEMMs of conv2/conv3/dense on bf16 MFMA, fp32 accumulate;
                                 0 = the reference's fp32 everywhere.  Nature-CNN only.  Passes of <= 512 frames without a backward (the actor step, the
                                 bootstrap row) stay on the fp32 small-batch kernels. */
  int32_t grad_accum_steps;   /* optax.MultiSteps every_k (ppo:79,492-500): each of the num_minibatches*k micro-batches feeds a running
                                 mean; the optimizer steps on every k-th.  0/1 = off. */
  int32_t async_batch_size;   /* legacy `--async-batch-size` (naturecnn:65-66): envpool returns this many of the local_num_envs envs per
                                 recv(); a rollout is num_steps * local_num_envs/async_batch_size rows of async_batch_size samples with env
                                 ids, returns are env-id-indexed (naturecnn:467-531), advantages normalised per minibatch
                                 (naturecnn:540-541).  PPO, one actor slot.  0 = synchronous (cleanba_ppo.py). */
  int32_t backward_split;     /* build-only extension, default 0 = the backward GEMMs on fp32 MFMA.  2 or 3: every fp32 operand of the
                                 backward GEMMs is split exactly into that many bf16 terms and the products are formed on bf16 MFMA with
                                 fp32 accumulation (2: a1b1+a1b2+a2b1, product error ~2^-16; 3: six products, ~2^-22 = fp32 rounding
                                 noise).  The forward pass, losses, returns and the optimizer are untouched.  Nature-CNN only. */
  int32_t num_channels;       /* Network(channels, hiddens) of ppo:92-95,175-176 (`--channels`, `--hiddens`).  channels: the IMPALA-ResNet torso is  */
  int32_t channels[4];        /* built for the reference default (16, 32, 32) and cbm_ctx_create REJECTS anything else rather than silently training */
  int32_t num_hiddens;        /* a different network; hiddens: ONE hidden layer of 64, 128, ... 512 units (default 256; flat parameter layout and    */
  int32_t hiddens[4];         /* cbm_param_count_hidden follow it).  Both ignored for CBM_NET_NATURE (the legacy script has no such flags).          */
  int32_t conv1_fp32_chain;   /* Nature-CNN, passes of more than 512 frames (the learner's minibatches).  0 (default): conv1 forward and weight
                                 gradient form EXACT products on the bf16 matrix cores — a pixel (integer 0..255) is exact in bf16, the fp32 operand
                                 (w/255, dY) is split into three 8-bit terms that sum to it exactly, accumulation in fp32: no operand bit is dropped,
                                 results differ from the fp32 chain by the order of roundings only (logits / values within 1e-5 of the oracle, measured
                                 1e-6; tests/test_gpu_conv1_exact.py).  Bit 0 set: the forward as the k-ascending fp32 fmaf chain on
                                 v_mfma_f32_32x32x2_f32, bit-identical to the oracle (what rounds 1-5 shipped); bit 1 set: the weight gradient on fp32
                                 MFMA; 3 = both.  The actor step (<= 512 frames) always uses the chain: sampled actions stay bit-exact either way. */
  int32_t reserved[2];
} cbm_config;

/* Fills cfg with the reference defaults for `algo` (ppo:34-118 / impala:34-110). */
int cbm_default_config(int32_t algo, cbm_config* cfg);
int32_t cbm_config_size(void);   /* sizeof(cbm_config) as the library was compiled: an FFI binding checks its own struct against it */

int cbm_ctx_create(const cbm_config* cfg, cbm_ctx** out);
int cbm_ctx_destroy(cbm_ctx* ctx);
const char* cbm_last_error(void);
const char* cbm_build_info(void);

/* ---- parameters: flat fp32 blob in flax layout (SURVEY §5 checkpoint names).
 * Replaces network.init/actor.init/critic.init + device_put (ppo:481-502) and the
 * params hand-off to the actors (ppo:721-725). */
int64_t cbm_param_count(int32_t network, int32_t num_actions);
int64_t cbm_param_count_hidden(int32_t network, int32_t num_actions, int32_t hidden);   /* IMPALA-ResNet with Network(hiddens=(hidden,)), ppo:94 */
int cbm_params_set(cbm_ctx* ctx, const float* host_params, int64_t n);      /* learner + actor copy, resets optimizer state */
int cbm_params_get(cbm_ctx* ctx, float* host_params, int64_t n);            /* learner copy */
int cbm_actor_params_get(cbm_ctx* ctx, float* host_params, int64_t n);      /* actor copy (policy version behind) */

/* ---- named device buffers (for tests, all-reduce plumbing, checkpointing).
 * names: "params", "actor_params", "actor_params_latest", "grads", "opt_m", "opt_v", "adv", "target", "perm",
 *        "obs", "actions", "logprobs", "values", "rewards", "dones", "env_ids", "logits", "stats",
 *        "actor_params_v0" .. "actor_params_v2" (the versioned actor copies) ...   */
int cbm_buffer(cbm_ctx* ctx, const char* name, int32_t ring_index, void** dev_ptr, int64_t* nbytes);
int cbm_copy_to_host(cbm_ctx* ctx, void* host_dst, const void* dev_src, int64_t nbytes);
int cbm_copy_to_device(cbm_ctx* ctx, void* dev_dst, const void* host_src, int64_t nbytes);
int cbm_dev_alloc(int64_t nbytes, void** dev_ptr);   /* plain hipMalloc/hipFree, so tests need no torch */
int cbm_dev_free(void* dev_ptr);
void* cbm_learner_stream(cbm_ctx* ctx);   /* hipStream_t, so torch.distributed can order its all-reduce */
int cbm_sync(cbm_ctx* ctx);

/* ---- actor side: replaces rollout()'s hot loop, ppo:308-375 / impala:351-416.
 * PRNG key handling = jax.random.split per step (ppo:256): the slot's key lives in the ctx. */
int cbm_actor_set_key(cbm_ctx* ctx, int32_t slot, const uint32_t key[2]);          /* ppo:677 */
int cbm_actor_get_key(cbm_ctx* ctx, int32_t slot, uint32_t key[2]);
/* Blocks until the slot may start rollout `update` (ring entry free + params published);
 * implements params_queue.get() and the `update != 2` skew (ppo:287-304).  Returns the
 * actor policy version in *policy_version. */
int cbm_actor_begin_rollout(cbm_ctx* ctx, int32_t slot, int32_t concurrency, int32_t* policy_version);
/* Host-env step (envpool numpy arrays): uploads obs[E,4,84,84] + done[E] into ring row t,
 * runs get_action_and_value (ppo:246-261), returns actions[E] after the 4*E-byte D2H (ppo:317). */
int cbm_actor_step_host(cbm_ctx* ctx, int32_t slot, const uint8_t* obs, const uint8_t* done,
                        const uint8_t* firststep, const float* reward_with_obs, int32_t* actions_out);
/* Async host-env step (legacy script, naturecnn:346-367): one envpool.recv() batch — obs[Ba,4,84,84], the reward[Ba] / done[Ba] that
 * arrived with it, env_id[Ba] — is stored in ring row t, get_action_and_value runs on it, actions_out[Ba] feed envs.send(action, env_id).
 * A rollout is exactly num_steps*async_update such calls followed by cbm_actor_commit(ctx, slot, NULL, NULL). */
int cbm_actor_step_async(cbm_ctx* ctx, int32_t slot, const uint8_t* obs, const float* reward, const uint8_t* done, const int32_t* env_id,
                         int32_t* actions_out);
/* reward_with_obs: IMPALA stores the reward that arrived WITH obs_t (impala:372-384); NULL for PPO,
 * which instead records what envs.step returned for the action just taken (ppo:321-342): */
int cbm_actor_record_host(cbm_ctx* ctx, int32_t slot, const float* reward);
/* Page-lock a host buffer the env hands to cbm_actor_step_host / cbm_actor_commit again and again (envpool's recv buffers, a numpy array
 * the host reuses): the 3.39 MB observation upload of a 120-env step (the jnp transfer inside get_action_and_value, ppo:313) then goes by
 * DMA from the caller's pages instead of through the runtime's pageable staging.  Optional; unregister before freeing the memory. */
int cbm_host_register(cbm_ctx* ctx, void* ptr, int64_t nbytes);
int cbm_host_unregister(cbm_ctx* ctx, void* ptr);
/* Device-env rollout: the built-in synthetic Atari-shaped env (cbm_synth_*) steps on the GPU, so
 * the whole T-step rollout is enqueued without host round trips.  nsteps = T (PPO) / T or T+1 (IMPALA). */
int cbm_actor_rollout_device(cbm_ctx* ctx, int32_t slot, int32_t nsteps);
/* Publishes the rollout (ppo:357-375): next_obs/next_done from the host env, or NULL for the
 * device env.  Bumps the ring sequence; never blocks on the learner. */
int cbm_actor_commit(cbm_ctx* ctx, int32_t slot, const uint8_t* next_obs, const uint8_t* next_done);
/* Episode statistics kept on the device env (ppo:343-352): mean returned episodic return/length. */
int cbm_actor_episode_stats(cbm_ctx* ctx, int32_t slot, float* avg_return, float* avg_length);

/* ---- split topologies (actor GPU != learner GPUs, ppo:97-100 / README.md:62 `--actor-device-ids 0 --learner-device-ids 1 2 3`):
 * replaces jax.device_put_sharded of the rollout shards (ppo:358-363) and the cross-device params put (ppo:721-725)
 * when actor and learners live in different processes.  Learner side: slots act as ingest ports. */
int cbm_ingest_begin(cbm_ctx* ctx, int32_t slot, int32_t* ring_index);   /* blocks until a ring entry is free; caller then fills cbm_buffer(name, ring_index) */
int cbm_ingest_commit(cbm_ctx* ctx, int32_t slot);                        /* publishes the entry to cbm_learner_wait; the caller's copies into it must have completed */
int cbm_params_publish_external(cbm_ctx* ctx, const float* dev_params, int64_t n);  /* actor side: a new parameter version arrived */
void* cbm_actor_stream(cbm_ctx* ctx, int32_t slot);                       /* hipStream_t of an actor slot (ordering of shard sends) */
int cbm_actor_ring_index(cbm_ctx* ctx, int32_t slot);                     /* ring entry of the slot's current / last rollout */

/* ---- collectives among the learner GPUs: RCCL over xGMI behind the C ABI.  Replaces jax.pmap(axis_name="local_devices",
 * devices=global_learner_decices) + jax.lax.pmean (ppo:628,649-660 / impala:621,636-645).  librccl is bound at run time (dlopen); the host
 * carries the 128-byte unique id from the rank that made it to the others (any transport: a file, a TCP store, MPI ...). */
#define CBM_COMM_SLOTS 2
#define CBM_COMM_LEARNERS 0        /* every learner rank of every actor-learner group: gradients + loss statistics */
#define CBM_COMM_WORLD 1           /* spare slot (job-wide barriers when actors take part) */
#define CBM_COMM_ID_BYTES 128
int cbm_comm_load(const char* librccl_path);                 /* optional: which librccl to bind (NULL = $CBM_RCCL_PATH, then the loader's) */
int cbm_comm_unique_id(uint8_t id[CBM_COMM_ID_BYTES]);
int cbm_comm_init(cbm_ctx* ctx, int32_t which, const uint8_t id[CBM_COMM_ID_BYTES], int32_t nranks, int32_t rank);
/* Self-test communicator: behaves like `nranks` ranks holding IDENTICAL data (all-reduce(SUM) = multiply by nranks, on the same streams and
 * behind the same events as the RCCL path), so stream-ordering bugs of the overlapped all-reduce show up as changed bits on one GPU. */
int cbm_comm_init_loopback(cbm_ctx* ctx, int32_t which, int32_t nranks);
/* Native backend ("the native design"): the same collectives WITHOUT RCCL — one HIP kernel per all-reduce working straight
 * on the peers' buffers through HIP IPC mappings.  Two-shot for the flat gradient: rank r reduces slice r of every peer's buffer by PEER READS,
 * summing in rank order 0..n-1 (deterministic: ppo:30's XLA_FLAGS asks the same of the reference), and writes the result into slice r of every
 * peer's buffer by PEER WRITES; arrival / completion flags live in a per-rank signal block the peers store to with system-scope atomics.
 * One-shot (every rank reads all, keeps the result in registers until all have read) for the statistics and the f64 scratch.  Ranks may be
 * processes on DIFFERENT GPUs of an xGMI node or several processes on ONE GPU (which RCCL refuses) — the way BASELINE configs[3] is tested on
 * a one-GPU box.  Set-up: every rank exports its blob, the host carries all blobs to every rank (any transport), every rank inits with the table.
 * A flag wait that exceeds $CBM_NATIVE_TIMEOUT_S (default 120) — a dead peer — makes the next host-synchronising call return an error. */
#define CBM_NATIVE_MAX_RANKS 16
#define CBM_NATIVE_BLOB_BYTES 320
int cbm_comm_native_export(cbm_ctx* ctx, int32_t which, uint8_t blob[CBM_NATIVE_BLOB_BYTES]);
int cbm_comm_native_init(cbm_ctx* ctx, int32_t which, int32_t nranks, int32_t rank, const uint8_t* blobs /* [nranks][CBM_NATIVE_BLOB_BYTES] */);
/* "rccl" | "native" | "loopback" | "" (slot not initialised) */
const char* cbm_comm_backend(cbm_ctx* ctx, int32_t which);
int cbm_comm_size(cbm_ctx* ctx, int32_t which);              /* ranks of an initialised communicator, 0 otherwise */
int cbm_comm_allreduce_f64(cbm_ctx* ctx, int32_t which, double* host_inout, int32_t n, int32_t op);   /* op: 0 sum, 1 max, 2 min; blocking */
int cbm_comm_barrier(cbm_ctx* ctx, int32_t which);
int cbm_comm_allreduce_grads(cbm_ctx* ctx, int32_t which);   /* all-reduce(SUM) of the whole flat gradient through communicator `which`, blocking (backend A/B on identical data) */
/* the learner-stream cost of a concurrent all-reduce (pmean under value_and_grad's backward, ppo:619-628): `iters` backward passes of one learner
 * minibatch alone, then each beside ONE whole-gradient all-reduce through `which` on the communication stream.
 * out = {ms per backward alone, ms per backward beside the all-reduce, us per all-reduce beside the backward}; overwrites the gradient buffer */
int cbm_comm_overlap_probe(cbm_ctx* ctx, int32_t which, int32_t iters, double out[3]);
/* pmean of the flat gradient after cbm_learner_minibatch_grad (ppo:628): all-reduce(SUM) over CBM_COMM_LEARNERS on the library's
 * communication stream — the dense + heads tail under the conv backward, the head after it — and the learner stream joins.
 * *grad_div = rank count, to be passed to cbm_learner_optimizer_step / cbm_learner_accumulate.  Without a communicator: no work,
 * *grad_div = 1.  cbm_learner_update does exactly this internally, so a data-parallel host makes the same single call as a one-GPU host;
 * cbm_learner_finish all-reduces the loss statistics (pmean, ppo:649-653) when stats are requested (every rank must then request them). */
int cbm_learner_allreduce_grads(cbm_ctx* ctx, float* grad_div);
int cbm_comm_profile(cbm_ctx* ctx, int32_t on);              /* HIP-event timing of every gradient all-reduce from now on */
int cbm_comm_profile_read(cbm_ctx* ctx, double* tail_ms, double* exposed_ms, int32_t* count);   /* totals: tail all-reduce; backward end -> optimizer may start */

/* ---- peer writes for split topologies: the actor writes rollout shards straight into the learners' rings, learner 0 writes parameters
 * straight into the actor's version buffers (HIP IPC mappings, strided copies on a side stream).  Replaces jax.device_put_sharded
 * (ppo:358-363) and jax.device_put(params, actor device) (ppo:721-725) between processes. */
#define CBM_IPC_HANDLE_BYTES 64
#define CBM_IPC_WINDOW_BYTES 128
typedef struct {   /* device pointers of ONE ring entry of the destination context (mapped window base + cbm_ipc_window_offset, or cbm_buffer of a local ctx); NULL = skip */
  void *obs, *actions, *logprobs, *values, *rewards, *dones, *firststeps, *logits;
} cbm_peer_ring;
/* EXPORT WINDOWS.  Everything another process may map lives in one device allocation per context and purpose — window 0: every ring entry's
 * fields and the three versioned actor parameter buffers; window 1: the flat gradient, the loss statistics and the f64 scratch (what the native
 * all-reduce works on; fine-grained memory when CBM_COMM=native is to span devices) — so a peer maps ONE HIP IPC handle per context pair and
 * addresses fields by offset.  The owner publishes the window's blob (handle + size + its pid / GPU / `tag`, an integer of the host's choosing —
 * its rank — that error messages quote) and the offsets of the fields; a peer opens the blob once (opening the same window again in the same
 * process returns the same mapping, reference-counted; a failed hipIpcOpenMemHandle is retried and every failure names both sides on stderr).
 * TEARDOWN ORDER of a multi-process host: cbm_ipc_close_all on every process -> a host barrier -> cbm_ctx_destroy; an owner must not free a
 * window while a peer still maps it. */
int cbm_ipc_export_window(cbm_ctx* ctx, int32_t window, int32_t tag, uint8_t blob[CBM_IPC_WINDOW_BYTES]);
int cbm_ipc_window_offset(cbm_ctx* ctx, const char* name /* a cbm_buffer name */, int32_t ring_index, int32_t* window, int64_t* offset, int64_t* nbytes);
int cbm_ipc_open_window(cbm_ctx* ctx, const uint8_t blob[CBM_IPC_WINDOW_BYTES], const char* what /* for error messages, may be NULL */, void** base);
int cbm_ipc_close_window(cbm_ctx* ctx, void* base);
int cbm_ipc_close_all(cbm_ctx* ctx);   /* every window this context opened + the native communicators' peer mappings (those communicators end here) */
/* Enqueues, on the io stream and after the commit of `slot`'s rollout in ring entry `ring_index`, the copy of learner `li`'s column shard
 * (columns [slot*E + li*E/L, +E/L) of every [T+1][B] field) into columns [dst_col0, +E/L) of `dst`, whose rows have dst_cols columns. */
int cbm_actor_ship_shard(cbm_ctx* ctx, int32_t slot, int32_t ring_index, int32_t li, int32_t n_learners, const cbm_peer_ring* dst,
                         int32_t dst_cols, int32_t dst_col0);
int cbm_io_sync(cbm_ctx* ctx);                                /* everything enqueued on the io stream has landed */
int cbm_params_push(cbm_ctx* ctx, void* const peer_versions[3]);   /* learner 0: current parameters -> the actor's version buffer; blocking */
int cbm_params_mark_published(cbm_ctx* ctx);                  /* actor: the next parameter version has landed in its buffer */
/* Makes every blocking wait of the context (cbm_actor_begin_rollout, cbm_ingest_begin, cbm_learner_wait) return an error from now on:
 * the host calls it when one of its threads has failed, so the others do not hang on a queue that will never be fed. */
int cbm_ctx_abort(cbm_ctx* ctx);

/* ---- learner side: replaces multi_device_update (ppo:579-660 / impala:599-645).
 * cbm_learner_wait blocks until every slot has committed rollout #update (ppo:697-711). */
int cbm_learner_wait(cbm_ctx* ctx);
/* Whole single-GPU update: GAE + adv-norm + epochs x minibatches x (loss, grads, optimizer).
 * `key` is the learner PRNG key (ppo:470), updated in place exactly as jax.random.split would.
 * lrs[i] / bc1[i] / bc2[i] are the schedule values of optimizer step i (linear_schedule
 * ppo:475-479 evaluated by the host in float32).  stats_out: [epochs*minibatches][5]
 * (loss, pg, v, entropy, approx_kl) for PPO, [minibatches][4] for IMPALA. */
int cbm_learner_update(cbm_ctx* ctx, uint32_t key[2], const float* lrs, const float* bc1, const float* bc2,
                       int32_t n_opt_steps, float* stats_out);
/* Split form for data-parallel learners (pmean of grads, ppo:628): prepare -> per minibatch
 * grad -> [caller all-reduces "grads"] -> optimizer step -> finish. */
int cbm_learner_prepare(cbm_ctx* ctx, uint32_t key[2]);
int cbm_learner_epoch_begin(cbm_ctx* ctx, uint32_t key[2]);   /* key,subkey = split(key); perm = permutation(subkey) ppo:599-606 */
int cbm_learner_minibatch_grad(cbm_ctx* ctx, int32_t epoch, int32_t minibatch);
int64_t cbm_learner_grad_tail_offset(cbm_ctx* ctx);   /* first element of the dense + heads tail of the flat gradient (all-reduced under the conv backward) */
/* gradient accumulation (grad_accum_steps = k > 1), split form: after each micro-batch's all-reduce call cbm_learner_accumulate with
 * mini_step = micro_batch % k; it folds "grads"/grad_div into the running mean and, on mini_step == k-1, leaves that mean in "grads" for
 * cbm_learner_optimizer_step(..., grad_div = 1). */
int cbm_learner_accumulate(cbm_ctx* ctx, int32_t mini_step, float grad_div);
int cbm_learner_optimizer_step(cbm_ctx* ctx, float lr, float bc1, float bc2, float grad_div);
int cbm_learner_finish(cbm_ctx* ctx, float* stats_out);   /* publishes params to the actors (ppo:721-725) */

/* ---- pure-function entry points (device pointers; used by the parity tests) ------------- */
/* get_action_and_value on B frames: ppo:246-261.  outputs may be NULL. */
int cbm_forward(cbm_ctx* ctx, const float* params, const uint8_t* obs, const int32_t* idx, int32_t B,
                int32_t dense_ksplit, float* logits, float* value);
int cbm_sample(cbm_ctx* ctx, const float* logits, int32_t B, const uint32_t subkey[2], int32_t* actions, float* logprobs);
int cbm_gae(cbm_ctx* ctx, const float* rewards, const float* values, const uint8_t* dones, const float* next_value,
            const uint8_t* next_done, int32_t T, int32_t B, float* adv, float* target);             /* ppo:532-560 */
int cbm_advnorm(cbm_ctx* ctx, float* adv, int32_t T, int32_t B, int32_t groups);                  /* ppo:592-595 */
/* rlax.vtrace_td_error_and_advantage with lambda = 1 and all clip thresholds 1 (impala:559-567): inputs [T,B] (v_tm1 = V[:-1], v_t = V[1:],
 * rewards, discounts, rho = pi(a)/mu(a)), outputs [T,B]: errors (= stop-gradient targets minus v_tm1), pg advantages, q estimates. */
int cbm_vtrace(cbm_ctx* ctx, const float* v_tm1, const float* v_t, const float* r_t, const float* disc_t, const float* rho_tm1, int32_t T,
               int32_t B, float* errors, float* pg_adv, float* q_est);
/* async returns: env_ids/rewards/values [R,B] (reward and done as they arrived with each observation), dones u8 [R,B] -> adv, target [R,B]:
 * prepare_data's reward re-index naturecnn:232-255 + env-id-indexed compute_gae naturecnn:467-531 */
int cbm_gae_async(cbm_ctx* ctx, const int32_t* env_ids, const float* rewards, const float* values, const uint8_t* dones, int32_t R, int32_t B,
                  int32_t num_envs, float* adv, float* target);
/* out[idx[i]] = (adv[idx[i]] - mean) / (std + 1e-8) over the n samples of one minibatch (idx NULL = identity), naturecnn:540-541 */
int cbm_mb_advnorm(cbm_ctx* ctx, const float* adv, const int32_t* idx, int32_t n, float* out);
int cbm_permutation(cbm_ctx* ctx, const uint32_t key[2], int32_t n, int32_t* perm);               /* ppo:606 */
int cbm_ppo_loss_grad(cbm_ctx* ctx, const float* params, const uint8_t* obs, const int32_t* idx, int32_t N,
                      const int32_t* actions, const float* old_logprob, const float* adv, const float* target,
                      float* stats5, float* grads, float* logits_out, float* value_out);           /* ppo:562-577,619 */
int cbm_impala_loss_grad(cbm_ctx* ctx, const float* params, const uint8_t* obs, const int32_t* idx, int32_t T1,
                         int32_t Bm, const float* mu_logits, const int32_t* actions, const float* rewards,
                         const uint8_t* dones, const uint8_t* firststeps, float* stats4, float* grads); /* impala:569-597 */
int cbm_adam_step(cbm_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t n, float max_norm, float lr,
                  float bc1, float bc2, float grad_div);                                            /* ppo:492-500,629 */
int cbm_rmsprop_step(cbm_ctx* ctx, float* p, const float* g, float* nu, int64_t n, float max_norm, float lr,
                     float grad_div);                                                               /* impala:152-188 */

/* ---- per-kernel HIP-event timing for bench.py's roofline line: brackets every learner-stream launch of
 * the selected implicit-GEMM kernel (ids in DESIGN.md §kernels; -1 = off). */
#define CBM_PROFILE_ALL (-2)   /* every launch of every id */
#define CBM_PROFILE_PAUSE (-3) /* stop recording; launches recorded so far stay readable (bench.py times a sample of the region) */
#define CBM_PROFILE_IDS 12
int cbm_profile_select(cbm_ctx* ctx, int32_t kernel_id);
int cbm_profile_read(cbm_ctx* ctx, double* total_ms, int32_t* count);
int cbm_profile_read_all(cbm_ctx* ctx, double* total_ms, int32_t* count, int32_t n_ids);   /* per-id totals after cbm_profile_select(CBM_PROFILE_ALL) */
/* the kernel that was LAUNCHED for `kernel_id` the last time it was timed, as "<kernel symbol> <problem functor type>" (empty string: never
 * timed).  bench.py holds profiles/pmc_traffic.json's per-kernel HBM traffic to this name and reports null when they differ. */
int cbm_profile_kernel_name(cbm_ctx* ctx, int32_t kernel_id, char* buf, int32_t buf_len);

/* ---- synthetic Atari-shaped environment (stands in for envpool.make, ppo:128-139) ------- */
typedef struct {
  int32_t elapsed, needs_reset;
  int32_t paddle_x, ball_x, ball_y, ball_dx, ball_dy;
  uint32_t bricks[3];
  uint32_t episode;
  float ep_return, ep_length, ret_return, ret_length;
  int32_t game;   /* 0 = Breakout preset; env e of an "Atari-57 mix" plays game e % 57 (BASELINE configs[4]) */
} cbm_env_state;
/* Host twin of the device env (same code path compiled for the CPU): steps n envs.
 * obs: [n,4,84,84] in/out frame stacks; outputs per env. */
int cbm_synth_env_reset_host(uint32_t seed, int32_t n, cbm_env_state* st, uint8_t* obs);
int cbm_synth_env_reset_host_games(uint32_t seed, int32_t n, int32_t atari57_mix, cbm_env_state* st, uint8_t* obs);
int cbm_synth_env_step_host(uint32_t seed, int32_t n, int32_t max_episode_steps, const int32_t* actions,
                            cbm_env_state* st, uint8_t* obs, float* reward, uint8_t* done, uint8_t* terminated,
                            int32_t* elapsed_step);
/* The same step out of place: reads the stacks in obs_prev, writes the new stacks to obs_next (a fresh array per step like envpool's
 * recv() hands out, without a second pass to copy it). */
int cbm_synth_env_step_host_to(uint32_t seed, int32_t n, int32_t max_episode_steps, const int32_t* actions, cbm_env_state* st,
                               const uint8_t* obs_prev, uint8_t* obs_next, float* reward, uint8_t* done, uint8_t* terminated,
                               int32_t* elapsed_step);
/* Diagnostics: paints the newest 84x84 plane of one env state, layered = 0 with the per-pixel function of the device kernels, 1 with the host
 * twin's layered painter (the two must agree byte for byte). */
int cbm_synth_env_render_host(const cbm_env_state* st, int32_t layered, uint8_t* plane);
/* envpool async mode, send(action, env_id) for a subset (impala:365, naturecnn:358): steps the k listed envs of the num_envs held in st / obs;
 * outputs in list order. */
int cbm_synth_env_step_host_ids(uint32_t seed, int32_t num_envs, int32_t k, int32_t max_episode_steps, const int32_t* env_ids,
                                const int32_t* actions, cbm_env_state* st, uint8_t* obs, float* reward, uint8_t* done,
                                uint8_t* terminated, int32_t* elapsed_step);
int cbm_actor_env_reset_device(cbm_ctx* ctx, int32_t slot, uint32_t seed);
int cbm_actor_env_reset_device_games(cbm_ctx* ctx, int32_t slot, uint32_t seed, int32_t atari57_mix);

#ifdef __cplusplus
}
#endif
#endif /* CLEANBA_MI_H */
