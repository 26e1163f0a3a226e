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

#define PADDLE_W 12
#define NBRICK 84  // 6 rows x 14 columns, 28 bits per state word

CBM_HD uint32_t env_hash(uint32_t seed, uint32_t env_id, uint32_t a, uint32_t b) {
  uint32_t o0, o1;
  cbm_threefry2x32(seed, env_id, a, b, &o0, &o1);
  return o0 ^ (o1 >> 3);
}

// "Atari-57 synthetic frame mix" (BASELINE configs[4]): env e plays game e % 57.  A game is a preset of the same
// pure-integer dynamics: action-set size (policy head stays 18 wide: action % n_actions, as envpool games with smaller action
// sets would ignore the rest), paddle width, ball speed, which brick rows exist, reward / termination rates (episode length) and
// a few static rectangles (frame sparsity).  Game 0 is the Breakout preset = the constants this env always had.
struct EnvGame { int32_t n_actions, paddle_w, speed_x, speed_y, rows_mask, reward_thr, term_thr, n_rects; uint32_t rect_key; };
CBM_HD EnvGame env_game(int32_t g) {
  EnvGame m;
  if (g == 0) { m.n_actions = 18; m.paddle_w = PADDLE_W; m.speed_x = 2; m.speed_y = 2; m.rows_mask = 0x3F; m.reward_thr = 1311; m.term_thr = 82;
                m.n_rects = 0; m.rect_key = 0; return m; }
  uint32_t a, b;
  cbm_threefry2x32(0xA7A5157u, (uint32_t)g, 57u, 0x51ED270Bu, &a, &b);
  m.n_actions = 4 + (int32_t)(a % 15u);                    // 4..18
  m.paddle_w = 8 + 2 * (int32_t)((a >> 4) & 7u);           // 8..22
  m.speed_x = 1 + (int32_t)((a >> 8) % 3u);
  m.speed_y = 1 + (int32_t)((a >> 12) % 3u);
  m.rows_mask = (int32_t)((a >> 16) & 0x3Fu);              // any subset of the six brick rows
  m.reward_thr = 328 + (int32_t)((a >> 22) % 2949u);       // p(reward) in [0.005, 0.05)
  m.term_thr = 22 + (int32_t)(b % 197u);                   // p(termination) in [1/3000, 1/300)
  m.n_rects = (int32_t)((b >> 8) % 6u);
  m.rect_key = b >> 11;
  return m;
}

CBM_HD void env_new_episode(cbm_env_state* s, uint32_t seed, uint32_t env_id) {
  const EnvGame gm = env_game(s->game);
  s->episode += 1u;
  const uint32_t h = env_hash(seed, env_id, s->episode, 0x9E3779B9u);
  s->elapsed = 0;
  s->needs_reset = 0;
  s->paddle_x = 36;
  s->ball_x = 4 + (int32_t)(h % 72u);
  s->ball_y = 40;
  s->ball_dx = (h >> 8) & 1u ? gm.speed_x : -gm.speed_x;
  s->ball_dy = gm.speed_y;
  s->bricks[0] = s->bricks[1] = s->bricks[2] = 0x0FFFFFFFu;
}

// one env.step(action); returns clipped reward, sets *terminated / *truncated
CBM_HD float env_advance(cbm_env_state* s, uint32_t seed, uint32_t env_id, int32_t action, int32_t max_steps, int* terminated,
                         int* truncated) {
  const EnvGame gm = env_game(s->game);
  s->elapsed += 1;
  const int dir = (action % gm.n_actions) % 3;
  int px = s->paddle_x + (dir == 1 ? 4 : (dir == 2 ? -4 : 0));
  s->paddle_x = px < 1 ? 1 : (px > 83 - gm.paddle_w ? 83 - gm.paddle_w : px);
  int bx = s->ball_x + s->ball_dx, by = s->ball_y + s->ball_dy;
  if (bx < 1) { bx = 1; s->ball_dx = -s->ball_dx; }
  if (bx > 81) { bx = 81; s->ball_dx = -s->ball_dx; }
  if (by < 12) { by = 12; s->ball_dy = -s->ball_dy; }
  if (by > 75) { by = 75; s->ball_dy = -s->ball_dy; }
  s->ball_x = bx; s->ball_y = by;
  // events depend on the action through the paddle position: nothing can be precomputed
  const uint32_t h = env_hash(seed ^ (s->episode * 0x85EBCA6Bu), env_id, (uint32_t)s->elapsed, (uint32_t)s->paddle_x);
  float reward = 0.0f;
  if ((h & 0xFFFFu) < (uint32_t)gm.reward_thr) {  // Breakout preset: ~0.02
    reward = 1.0f;
    uint32_t k = (h >> 7) % NBRICK;
    for (int tries = 0; tries < NBRICK; ++tries) {  // clear the next standing brick
      const uint32_t w = k / 28u, bit = k % 28u;
      if (s->bricks[w] & (1u << bit)) { s->bricks[w] &= ~(1u << bit); break; }
      k = (k + 1u) % NBRICK;
    }
    if ((s->bricks[0] | s->bricks[1] | s->bricks[2]) == 0u) s->bricks[0] = s->bricks[1] = s->bricks[2] = 0x0FFFFFFFu;
  }
  *terminated = ((h >> 16) & 0xFFFFu) < (uint32_t)gm.term_thr ? 1 : 0;  // Breakout preset: ~1/800
  *truncated = s->elapsed >= max_steps ? 1 : 0;
  return reward;
}

CBM_HD uint8_t env_pixel(const cbm_env_state* s, const EnvGame& gm, int y, int x) {
  if (y >= 17 && y < 35) {  // six brick rows, 2 px tall + 1 px gap; 14 bricks of 5 px + 1 px gap
    const int row = (y - 17) / 3, ry = (y - 17) % 3, col = x / 6, rx = x % 6;
    if (ry < 2 && rx < 5 && ((gm.rows_mask >> row) & 1)) {
      const int k = row * 14 + col;
      if (s->bricks[k / 28] & (1u << (k % 28))) return (uint8_t)(200 - 24 * row);
    }
    return 0;
  }
  if (y >= 78 && y < 80 && x >= s->paddle_x && x < s->paddle_x + gm.paddle_w) return 200;
  if (y >= s->ball_y && y < s->ball_y + 2 && x >= s->ball_x && x < s->ball_x + 2) return 255;
  if (y >= 10 && y < 12) return 142;
  if (y >= 12 && (x == 0 || x == 83)) return 142;
  for (int r = 0; r < gm.n_rects; ++r) {   // static scenery of the game preset (rows 36..75: below the bricks, above the paddle)
    const uint32_t k = gm.rect_key * 2654435761u + (uint32_t)r * 0x9E3779B9u;
    const int ry0 = 36 + (int)(k % 32u), rx0 = 2 + (int)((k >> 5) % 64u), rh = 2 + (int)((k >> 11) % 6u), rw = 4 + (int)((k >> 14) % 14u);
    if (y >= ry0 && y < ry0 + rh && x >= rx0 && x < rx0 + rw && x < 83) return (uint8_t)(90 + 20 * r);
  }
  return 0;
}

// full transition of one env given its previous frame stack; pixel work done by the caller's threads
struct EnvOut { float reward; uint8_t done, terminated, firststep, was_reset; int32_t elapsed; };

CBM_HD EnvOut env_transition(cbm_env_state* s, uint32_t seed, uint32_t env_id, int32_t action, int32_t max_steps) {
  EnvOut o;
  if (s->needs_reset) {
    env_new_episode(s, seed, env_id);
    o.reward = 0.0f; o.done = 0; o.terminated = 0; o.firststep = 1; o.was_reset = 1; o.elapsed = 0;
    return o;
  }
  int term = 0, trunc = 0;
  o.reward = env_advance(s, seed, env_id, action, max_steps, &term, &trunc);
  o.terminated = (uint8_t)term;
  o.done = (uint8_t)(term | trunc);
  o.firststep = 0; o.was_reset = 0; o.elapsed = s->elapsed;
  s->ep_return += o.reward;
  s->ep_length += 1.0f;
  if (o.done) {
    s->ret_return = s->ep_return; s->ret_length = s->ep_length;
    s->ep_return = 0.0f; s->ep_length = 0.0f;
    s->needs_reset = 1;
  }
  return o;
}

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
  __shared__ cbm_env_state s;
  __shared__ EnvOut out;
  __shared__ EnvGame gm;
  const int e = blockIdx.x;
  if (threadIdx.x == 0) {
    s = st[e];
    gm = env_game(s.game);
    out = env_transition(&s, seed, (uint32_t)e, actions[e], max_steps);
    st[e] = s;
    reward[e] = out.reward;
    done_next[e] = out.done;
    if (firststep_next) firststep_next[e] = out.firststep;
  }
  __syncthreads();
  const uint8_t* p = obs_prev + (size_t)e * CBM_FRAME;
  uint8_t* o = obs_next + (size_t)e * CBM_FRAME;
  const bool rs = out.was_reset;
  // 4-byte granularity: 1764 words per plane
  const uint32_t* p32 = reinterpret_cast<const uint32_t*>(p);
  uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
  // the three older planes are shifted from the previous stack: all of a thread's loads are issued before its first store (the loop below
  // used to pay one L2 round trip per iteration, seven in a row, on the critical path of every env-step)
  uint32_t older[7][3];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int i = threadIdx.x + 256 * j;
    if (i < 1764 && !rs) { older[j][0] = p32[1764 + i]; older[j][1] = p32[2 * 1764 + i]; older[j][2] = p32[3 * 1764 + i]; }
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int i = threadIdx.x + 256 * j;
    if (i >= 1764) break;
    const int y = (4 * i) / 84, x = (4 * i) % 84;
    const uint32_t nw = (uint32_t)env_pixel(&s, gm, y, x) | ((uint32_t)env_pixel(&s, gm, y, x + 1) << 8) |
                        ((uint32_t)env_pixel(&s, gm, y, x + 2) << 16) | ((uint32_t)env_pixel(&s, gm, y, x + 3) << 24);
    if (rs) { o32[i] = nw; o32[1764 + i] = nw; o32[2 * 1764 + i] = nw; }
    else { o32[i] = older[j][0]; o32[1764 + i] = older[j][1]; o32[2 * 1764 + i] = older[j][2]; }
    o32[3 * 1764 + i] = nw;
  }
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
// partition gives the same bytes).  CBM_ENV_THREADS overrides the thread count (default min(8, cores), at least 8 envs per thread).
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
    return n < 1 ? 1 : (n > 8 && !e ? 8 : n);
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
