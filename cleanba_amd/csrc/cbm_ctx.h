// cbm_ctx.h — the context object behind the opaque cbm_ctx handle (shared by api.hip and comm.hip).
#pragma once
#include "cbm_internal.h"
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <vector>

#define MAX_SLOTS 16
#define MAX_RING 4
#define NPV 3
#define CBM_COMM_SCRATCH 16
#define CBM_COMM_PROF_MAX 1024

#define CBM_NATIVE_BUFS 4          // grads | loss statistics | f64 scratch | signal block  (cbm_comm_native_export order)
#define CBM_WINDOWS 2              // export windows of a context: 0 = ring fields + versioned actor parameters, 1 = gradient + statistics + f64 scratch
struct CbmComm {
  void* comm = nullptr;   // ncclComm_t
  int nranks = 0, rank = -1;
  bool loopback = false;  // self-test: nranks identical ranks, all-reduce(SUM) = scale by nranks
  // native backend (comm.hip, "native xGMI all-reduce"): every rank's gradient / statistics / scratch buffers and signal block mapped into this
  // process (HIP IPC, or the plain pointer when the peer context lives in this process); one kernel per collective, flags in the signal blocks
  bool native = false;
  void* nat_peer[CBM_NATIVE_BUFS][CBM_NATIVE_MAX_RANKS] = {};
  void* nat_win[2][CBM_NATIVE_MAX_RANKS] = {};   // mappings this communicator holds of rank r's communication window / signal block (cbm_ipc_map; null: own or same process)
  uint32_t nat_seq = 0;          // collective sequence number: the flag value of the next call
  bool nat_closed = false;       // cbm_ipc_close_all unmapped this communicator: the slot cannot be initialised again (its signal block still holds the old
                                 // sequence numbers and sticky error words; a fresh context is the way to a fresh communicator)
  void* nat_sig_local = nullptr; // this rank's signal block (owned)
  int* nat_err = nullptr;        // page-locked host word: a flag wait timed out (a peer died) — checked when the host next synchronises
};

struct RingEntry {
  uint8_t* obs = nullptr;
  int32_t* actions = nullptr;
  float *logprobs = nullptr, *values = nullptr, *rewards = nullptr, *logits = nullptr;
  uint8_t *dones = nullptr, *firststeps = nullptr;
  int32_t* env_ids = nullptr;   // async rollouts: which env each sample of a row belongs to (naturecnn:355,367)
  hipEvent_t ready[MAX_SLOTS];
  hipEvent_t consumed;
};
struct Slot {
  hipStream_t stream = nullptr;
  NatureWs ws;
  uint32_t key[2] = {0, 0};
  int t = 0, rollout = 0, ring = 0, pver = 0;
  cbm_env_state* env_state = nullptr;
  uint32_t env_seed = 0;
  bool env_inited = false;
  float* stats_dev = nullptr;
  // page-locked staging ring for the small per-step host arrays of the envpool-API path (dones, firststeps, rewards, actions): a pageable
  // hipMemcpyAsync of 120-480 bytes costs the host ~8 us each (runtime staging + bookkeeping), from page-locked memory it is a plain enqueue
  uint8_t* pin = nullptr;
  size_t pin_stride = 0;
  uint32_t pin_cur = 0;   // unsigned: the ring index stays in range after the counter wraps
};
#define CBM_PIN_RING 8   // staging entries per slot: an entry is reused 8 enqueues later, long after its copy ran (every step ends in a stream sync)
struct cbm_ctx {
  cbm_config cfg;
  NatureLayout L;
  int E, S, Bdev, T, T1, A, MB, nmb, epochs;
  int asyncB = 0, NE = 0;   // legacy --async-batch-size: rows of asyncB samples drawn from NE envs (0 = synchronous)
  int64_t P;
  float *params = nullptr, *grads = nullptr, *opt_m = nullptr, *opt_v = nullptr;
  float* actor_params[NPV] = {nullptr, nullptr, nullptr};
  hipEvent_t params_ready[NPV];
  RingEntry ring[MAX_RING];
  Slot slots[MAX_SLOTS];
  hipStream_t lstream = nullptr;
  NatureWs lws;
  float* advn = nullptr;     // per-minibatch normalised advantages (async mode, naturecnn:540-541)
  float *adv = nullptr, *target = nullptr, *next_value = nullptr, *stats_dev = nullptr, *loss_partials = nullptr, *norm_partials = nullptr;
  int32_t *perm = nullptr, *perm_tmp = nullptr, *impala_idx = nullptr;   // perm: [epochs][T * Bdev] (row 0 alone when the epochs are permuted one by one)
  const int32_t* perm_cur = nullptr;   // the current epoch's row of perm
  bool perm_batched = false;           // perm / perm_tmp / ckeys are sized for every epoch's permutation at once (launch_permutations_batch)
  float* gacc = nullptr;   // MultiSteps running mean (grad_accum_steps > 1)
  int accum = 1, nmicro = 0;
  uint64_t* ckeys = nullptr;
  // Lock-free hand-off (replaces the two Queue(maxsize=1) per actor thread, ppo:662-686): sequence numbers are atomics, a waiter sleeps on
  // the `epoch` word (futex via std::atomic::wait) and every publisher bumps it after its store — no mutex on either side.
  std::atomic<int> committed[MAX_SLOTS];
  std::atomic<int> updates_done{0};
  std::atomic<uint32_t> epoch{0};
  int stat_rows = 0;
  CbmProf prof;
  hipEvent_t tail_ev = nullptr, ext_ev = nullptr;   // gradient-tail hand-off to the communication stream
  // ---- comm.hip: RCCL communicators, the communication stream of the gradient all-reduce, the io stream of shard / parameter peer writes
  CbmComm comms[CBM_COMM_SLOTS];
  hipStream_t cstream = nullptr, iostream = nullptr;
  hipEvent_t bwd_ev = nullptr;
  std::mutex io_mu;
  double* comm_scratch = nullptr;
  bool comm_overlap = true;
  bool comm_prof_on = false, comm_prof_created = false;
  int comm_prof_n = 0;
  hipEvent_t comm_prof_ev[4 * CBM_COMM_PROF_MAX];
  std::atomic<bool> aborted{false};   // cbm_ctx_abort: every blocking wait returns an error from now on
  // ---- export windows: everything another process may map lives in ONE allocation per purpose, so a peer holds one mapping per context pair
  // and addresses fields by offset (window 0: ring fields + actor parameter versions; window 1: flat gradient | loss statistics | f64 scratch,
  // fine-grained when a native communicator is to span devices)
  uint8_t* win[CBM_WINDOWS] = {nullptr, nullptr};
  size_t win_bytes[CBM_WINDOWS] = {0, 0};
  bool win_fine[CBM_WINDOWS] = {false, false};
  std::mutex maps_mu;
  std::vector<void*> maps;   // peer windows this context opened with cbm_ipc_open_window (closed by cbm_ipc_close_all / destroy)
};
// publish: store the sequence number (release), then wake the sleepers
static inline void cbm_publish(cbm_ctx* c, std::atomic<int>& var, int value) {
  var.store(value, std::memory_order_release);
  c->epoch.fetch_add(1, std::memory_order_release);
  c->epoch.notify_all();
}
// wait until pred() or abort; returns false on abort.  The epoch is read BEFORE the predicate, so a publish between the two wakes us.
template <class Pred>
static inline bool cbm_wait(cbm_ctx* c, Pred pred) {
  for (;;) {
    const uint32_t e = c->epoch.load(std::memory_order_acquire);
    if (c->aborted.load(std::memory_order_acquire)) return false;
    if (pred()) return true;
    c->epoch.wait(e, std::memory_order_acquire);
  }
}
int cbm_learner_allreduce_grads_impl(cbm_ctx* c, float* grad_div);
int cbm_learner_allreduce_stats_impl(cbm_ctx* c);
int cbm_comm_destroy_all(cbm_ctx* c);
int cbm_comm_check_native(cbm_ctx* c);   // -1 (with the error set) when a native collective's flag wait timed out
int cbm_ipc_close_all_impl(cbm_ctx* c);  // unmaps every peer window and native-communicator mapping the context holds (idempotent)

