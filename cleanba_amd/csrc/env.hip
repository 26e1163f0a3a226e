// env.hip — synthetic Atari-shaped vector environment (stands in for envpool.make(...), ppo:128-139).
//
// Breakout-shaped 84x84 frames, 4-frame stacks, pure-integer dynamics keyed by (seed, env_id, episode,
// step, action): the same functions are compiled for the GPU (device env: frames are rendered straight
// into the HBM rollout ring, no host round trip) and for the CPU (host env with envpool's numpy API), so
// the two produce identical bytes (tests/test_env.py).
//
// envpool semantics kept: auto-reset on the step after `done` (that step returns the first
// observation of the new episode with reward 0 and elapsed_step 0), `terminated` w.p. ~1/800 per step,
// clipped reward in {0,1} w.p. ~0.02, truncation at max_episode_steps (ppo:121-123,328).
#include "cbm_internal.h"
#include <stdlib.h>
#include <unistd.h>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>
#include <string.h>

#include "env_model.h"

// ------------------------------------------------------------------------------------------ device
__global__ __launch_bounds__(256) void env_reset_kernel(uint32_t seed, int E, int mix, cbm_env_state* st, uint8_t* obs, int64_t stride,
                                                         uint8_t* done, uint8_t* firststep) {
  __shared__ cbm_env_state s;
  __shared__ EnvGame gm;
  const int e = blockIdx.x;
  if (threadIdx.x == 0) {
    s = cbm_env_state();
    s.episode = 0;
    s.game = mix ? e % 57 : 0;
    gm = env_game(s.game);
    env_new_episode(&s, seed, (uint32_t)e);
    st[e] = s;
    if (done) done[e] = 0;
    if (firststep) firststep[e] = 1;
  }
  __syncthreads();
  uint8_t* o = obs + (size_t)e * stride;
  for (int i = threadIdx.x; i < 7056; i += 256) {
    const uint8_t v = env_pixel(&s, gm, i / 84, i % 84);
    o[i] = v; o[7056 + i] = v; o[2 * 7056 + i] = v; o[3 * 7056 + i] = v;
  }
}
void launch_env_reset(uint32_t seed, int E, int mix, cbm_env_state* st_dev, uint8_t* obs, int64_t obs_stride, uint8_t* done, uint8_t* firststep,
                      hipStream_t st) {
  hipLaunchKernelGGL(env_reset_kernel, dim3(E), dim3(256), 0, st, seed, E, mix, st_dev, obs, obs_stride, done, firststep);
}

__global__ __launch_bounds__(256) void env_step_kernel(uint32_t seed, int E, int max_steps, const int32_t* actions, cbm_env_state* st,
                                                        const uint8_t* obs_prev, uint8_t* obs_next, float* reward, uint8_t* done_next,
                                                        uint8_t* firststep_next) {
  const EnvStepArgs a{seed, max_steps, st, obs_prev, obs_next, reward, done_next, firststep_next};
  const int e = blockIdx.x, tid = threadIdx.x;
  __shared__ EnvShared sh;
  EnvPieces<256> pc;
  env_step_prefetch<256>(a, e, tid, pc);
  if (tid < 3) {
    uint32_t sw[ENV_STATE_WORDS];
    env_state_load_words(st, e, sw);
    env_step_candidates(a, e, sh, tid, sw);
  }
  __syncthreads();
  env_step_early<256>(a, e, sh, tid, pc);
  env_step_finish<256>(a, e, sh, env_action_dir(sh, actions[e]), tid, pc);
}
void launch_env_step(uint32_t seed, int E, int max_episode_steps, const int32_t* actions, cbm_env_state* st_dev, const uint8_t* obs_prev,
                     uint8_t* obs_next, float* reward, uint8_t* done_next, uint8_t* firststep_next, hipStream_t st) {
  hipLaunchKernelGGL(env_step_kernel, dim3(E), dim3(256), 0, st, seed, E, max_episode_steps, actions, st_dev, obs_prev, obs_next, reward,
                     done_next, firststep_next);
}

__global__ void env_stats_kernel(const cbm_env_state* st, int E, float* out2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float r = 0.0f, l = 0.0f;
  for (int e = 0; e < E; ++e) { r += st[e].ret_return; l += st[e].ret_length; }
  out2[0] = r / (float)E; out2[1] = l / (float)E;
}
void launch_env_stats(const cbm_env_state* st_dev, int E, float* out2, hipStream_t st) {
  hipLaunchKernelGGL(env_stats_kernel, dim3(1), dim3(64), 0, st, st_dev, E, out2);
}

// ------------------------------------------------------------------------------------------ host twin
extern "C" int cbm_synth_env_reset_host(uint32_t seed, int32_t n, cbm_env_state* st, uint8_t* obs) {
  return cbm_synth_env_reset_host_games(seed, n, 0, st, obs);
}
extern "C" int cbm_synth_env_reset_host_games(uint32_t seed, int32_t n, int32_t atari57_mix, cbm_env_state* st, uint8_t* obs) {
  for (int e = 0; e < n; ++e) {
    cbm_env_state s = cbm_env_state();
    s.game = atari57_mix ? e % 57 : 0;
    const EnvGame gm = env_game(s.game);
    env_new_episode(&s, seed, (uint32_t)e);
    st[e] = s;
    uint8_t* o = obs + (size_t)e * CBM_FRAME;
    for (int i = 0; i < 7056; ++i) {
      const uint8_t v = env_pixel(&s, gm, i / 84, i % 84);
      o[i] = v; o[7056 + i] = v; o[2 * 7056 + i] = v; o[3 * 7056 + i] = v;
    }
  }
  return 0;
}
// env_pixel for a whole 84x84 plane, painted layer by layer in reverse order of env_pixel's precedence (scenery < side walls < top bar <
// ball < paddle; rows 17..34 belong to the bricks alone) instead of 7056 calls of the per-pixel function: ~20 us -> ~1 us per env on a host
// core.  tests/test_env.py compares it with the device kernel (which calls env_pixel) byte for byte.
static void host_render_plane(const cbm_env_state* s, const EnvGame& gm, uint8_t* o) {
  memset(o, 0, 7056);
  for (int r = gm.n_rects - 1; r >= 0; --r) {
    const uint32_t k = gm.rect_key * 2654435761u + (uint32_t)r * 0x9E3779B9u;
    const int ry0 = 36 + (int)(k % 32u), rx0 = 2 + (int)((k >> 5) % 64u), rh = 2 + (int)((k >> 11) % 6u), rw = 4 + (int)((k >> 14) % 14u);
    const int x1 = rx0 + rw < 83 ? rx0 + rw : 83;
    for (int y = ry0; y < ry0 + rh && y < 84; ++y)
      if (x1 > rx0) memset(o + y * 84 + rx0, 90 + 20 * r, (size_t)(x1 - rx0));
  }
  for (int y = 12; y < 84; ++y) { o[y * 84] = 142; o[y * 84 + 83] = 142; }
  memset(o + 10 * 84, 142, 2 * 84);
  for (int y = s->ball_y < 0 ? 0 : s->ball_y; y < s->ball_y + 2 && y < 84; ++y)
    for (int x = s->ball_x < 0 ? 0 : s->ball_x; x < s->ball_x + 2 && x < 84; ++x) o[y * 84 + x] = 255;
  for (int y = 78; y < 80; ++y)
    for (int x = s->paddle_x < 0 ? 0 : s->paddle_x; x < s->paddle_x + gm.paddle_w && x < 84; ++x) o[y * 84 + x] = 200;
  for (int y = 17; y < 35; ++y)
    for (int x = 0; x < 84; ++x) o[y * 84 + x] = env_pixel(s, gm, y, x);
}
static void host_step_one(uint32_t seed, int e, int32_t action, int32_t max_episode_steps, cbm_env_state* st, uint8_t* obs, float* reward,
                          uint8_t* done, uint8_t* terminated, int32_t* elapsed_step, const uint8_t* obs_prev = nullptr) {
  cbm_env_state s = st[e];
  const EnvGame gm = env_game(s.game);
  const EnvOut out = env_transition(&s, seed, (uint32_t)e, action, max_episode_steps);
  st[e] = s;
  *reward = out.reward; *done = out.done; *terminated = out.terminated; *elapsed_step = out.elapsed;
  uint8_t* o = obs + (size_t)e * CBM_FRAME;
  if (!out.was_reset) memmove(o, (obs_prev ? obs_prev + (size_t)e * CBM_FRAME : o) + 7056, 3 * 7056);
  host_render_plane(&s, gm, o + 3 * 7056);
  if (out.was_reset) { memcpy(o, o + 3 * 7056, 7056); memcpy(o + 7056, o + 3 * 7056, 7056); memcpy(o + 2 * 7056, o + 3 * 7056, 7056); }
}
// envpool steps its envs on a C++ thread pool; the twin does the same so that host-env runs are not bound by one core: the k envs of a call
// are cut into contiguous chunks, one std::thread each (an env's trajectory depends only on its own id, seed and actions — any
// partition gives the same bytes).  CBM_ENV_THREADS overrides the thread count (default min(16, cores), at least 8 envs per thread: 120 envs step on 15 threads in 47 us, on 8 in 72).  Measured in round 5 and NOT taken (tools/host_loop_probe.py, profiles/r05_host_loop_probe.txt): 4 envs per thread / up to 32 threads — env step 101-186 us instead of 48-86 (every worker is woken every step); workers that spin ~0.5 ms before sleeping — two actor threads' pools then starve each other and the Python threads (env step 237-298 us).
// The workers are a persistent pool per calling thread (envpool keeps its worker threads too; spawning eight std::threads per step cost
// ~0.4 ms of a 120-env step): the caller publishes a generation number, every worker runs its chunk and counts down, the caller runs
// chunk 0 itself and waits for the count.  thread_local, so two actor threads stepping their own envs never share a pool.
struct HostPool {
  std::vector<std::thread> th;
  std::atomic<uint32_t> gen{0};
  std::atomic<int> left{0};
  std::atomic<bool> stop{false};
  std::function<void(int)> job;   // job(t) runs chunk t
  const pid_t owner = getpid();   // a fork()ed child inherits the object but not the threads: it steps serially
  explicit HostPool(int workers) {
    for (int t = 1; t <= workers; ++t)
      th.emplace_back([this, t] {
        uint32_t seen = 0;
        for (;;) {
          for (int spin = 0; spin < 2000 && gen.load(std::memory_order_acquire) == seen; ++spin) __builtin_ia32_pause();
          gen.wait(seen, std::memory_order_acquire);
          seen = gen.load(std::memory_order_acquire);
          if (stop.load(std::memory_order_acquire)) return;
          job(t);
          if (left.fetch_sub(1, std::memory_order_acq_rel) == 1) left.notify_one();
        }
      });
  }
  ~HostPool() {
    // in a fork()ed child the std::thread objects name threads that do not exist there: joining a stale tid can block for ever at exit
    if (owner != getpid()) { for (auto& x : th) x.detach(); return; }
    stop.store(true, std::memory_order_release);
    gen.fetch_add(1, std::memory_order_release);
    gen.notify_all();
    for (auto& x : th) x.join();
  }
  void run(int nt, std::function<void(int)> f) {   // nt - 1 <= th.size() worker chunks + chunk 0 on the caller
    job = std::move(f);
    left.store((int)th.size(), std::memory_order_release);
    gen.fetch_add(1, std::memory_order_release);
    gen.notify_all();
    job(0);
    for (int spin = 0; spin < 4000 && left.load(std::memory_order_acquire) != 0; ++spin) __builtin_ia32_pause();
    for (int l; (l = left.load(std::memory_order_acquire)) != 0;) left.wait(l, std::memory_order_acquire);
    (void)nt;
  }
};
template <class F>
static void host_parallel_for(int k, F body) {
  static const int max_threads = [] {
    const char* e = getenv("CBM_ENV_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : (n > 16 && !e ? 16 : n);
  }();
  const int nt = k / 8 < max_threads ? k / 8 : max_threads;
  if (nt <= 1) { for (int j = 0; j < k; ++j) body(j); return; }
  thread_local HostPool pool(max_threads - 1);
  if (pool.owner != getpid()) { for (int j = 0; j < k; ++j) body(j); return; }
  pool.run(nt, [&](int t) {
    if (t >= nt) return;
    const int lo = (int)((int64_t)k * t / nt), hi = (int)((int64_t)k * (t + 1) / nt);
    for (int j = lo; j < hi; ++j) body(j);
  });
}
extern "C" int cbm_synth_env_step_host(uint32_t seed, int32_t n, int32_t max_episode_steps, const int32_t* actions, cbm_env_state* st,
                                       uint8_t* obs, float* reward, uint8_t* done, uint8_t* terminated, int32_t* elapsed_step) {
  host_parallel_for(n, [=](int e) { host_step_one(seed, e, actions[e], max_episode_steps, st, obs, reward + e, done + e, terminated + e, elapsed_step + e); });
  return 0;
}
// diagnostics: one 84x84 plane of a state, painted by the per-pixel function the device kernels call (layered = 0) or by the host twin's
// layered painter (layered = 1); tests/test_env.py compares the two over random states of every game preset
extern "C" int cbm_synth_env_render_host(const cbm_env_state* st, int32_t layered, uint8_t* plane) {
  const EnvGame gm = env_game(st->game);
  if (layered == 2) {   // the device kernels' word painter (env_word)
    for (int i = 0; i < 1764; ++i) { const uint32_t w = env_word(st, gm, i); memcpy(plane + 4 * i, &w, 4); }
    return 0;
  }
  if (layered) { host_render_plane(st, gm, plane); return 0; }
  for (int i = 0; i < 7056; ++i) plane[i] = env_pixel(st, gm, i / 84, i % 84);
  return 0;
}
extern "C" int cbm_synth_env_step_host_to(uint32_t seed, int32_t n, int32_t max_episode_steps, const int32_t* actions, cbm_env_state* st,
                                          const uint8_t* obs_prev, uint8_t* obs_next, float* reward, uint8_t* done, uint8_t* terminated,
                                          int32_t* elapsed_step) {
  host_parallel_for(n, [=](int e) {
    host_step_one(seed, e, actions[e], max_episode_steps, st, obs_next, reward + e, done + e, terminated + e, elapsed_step + e, obs_prev);
  });
  return 0;
}
// envpool's send(action, env_id) for a subset: steps the k envs listed in env_ids (indices into st / obs, which hold ALL envs); the per-env
// outputs are written in list order.
extern "C" int cbm_synth_env_step_host_ids(uint32_t seed, int32_t num_envs, int32_t k, int32_t max_episode_steps, const int32_t* env_ids,
                                           const int32_t* actions, cbm_env_state* st, uint8_t* obs, float* reward, uint8_t* done,
                                           uint8_t* terminated, int32_t* elapsed_step) {
  for (int j = 0; j < k; ++j)
    if (env_ids[j] < 0 || env_ids[j] >= num_envs) { cbm_set_error("env_id %d outside [0,%d)", env_ids[j], num_envs); return -1; }
  host_parallel_for(k, [=](int j) {
    host_step_one(seed, env_ids[j], actions[j], max_episode_steps, st, obs, reward + j, done + j, terminated + j, elapsed_step + j);
  });
  return 0;
}
