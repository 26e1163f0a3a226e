// env_model.h — the synthetic env's pure-integer game model (shared by env.hip, which owns the kernels and the host twin, and by
// gemm_layers.hip, whose per-frame actor tail steps its env in the same launch that sampled the action).
#pragma once
#include "cbm_internal.h"

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


// the four pixels x .. x+3 of word i of the 84x84 plane (21 words per row, pixel x in byte x % 4): env_pixel's bytes, decided per REGION
// instead of per pixel — empty rows, top bar, side walls and brick rows directly, only the words that touch the paddle rows, the ball or a
// scenery rectangle through the per-pixel function (tests/test_env.py holds it to env_pixel over random states of all 57 presets)
CBM_HD uint32_t env_word(const cbm_env_state* s, const EnvGame& gm, int i) {
  const int y = i / 21, x = 4 * (i - 21 * y);
  if (y < 10) return 0u;
  if (y >= 17 && y < 35) {
    const int row = (y - 17) / 3, ry = (y - 17) - 3 * row;
    if (ry == 2 || !((gm.rows_mask >> row) & 1)) return 0u;
    const uint32_t val = (uint32_t)(200 - 24 * row);
    uint32_t w = 0u;
    for (int p = 0; p < 4; ++p) {
      const int xx = x + p, col = xx / 6, rx = xx - 6 * col, k = row * 14 + col;
      if (rx < 5 && ((s->bricks[k / 28] >> (k % 28)) & 1u)) w |= val << (8 * p);
    }
    return w;
  }
  bool slow = (y >= 78 && y < 80) || (y >= s->ball_y && y < s->ball_y + 2 && x + 3 >= s->ball_x && x < s->ball_x + 2);
  for (int r = 0; r < gm.n_rects; ++r) {
    const uint32_t k = gm.rect_key * 2654435761u + (uint32_t)r * 0x9E3779B9u;
    const int ry0 = 36 + (int)(k % 32u), rh = 2 + (int)((k >> 11) % 6u);
    slow = slow || (y >= ry0 && y < ry0 + rh);
  }
  if (slow)
    return (uint32_t)env_pixel(s, gm, y, x) | ((uint32_t)env_pixel(s, gm, y, x + 1) << 8) | ((uint32_t)env_pixel(s, gm, y, x + 2) << 16) |
           ((uint32_t)env_pixel(s, gm, y, x + 3) << 24);
  if (y < 12) return 0x8E8E8E8Eu;                                  // top bar (142)
  return (x == 0 ? 142u : 0u) | (x == 80 ? 142u << 24 : 0u);       // side walls at x = 0 and x = 83
}

// ---- one env's step by one 256-thread block, arranged so that almost nothing is left to do once the action exists (the per-frame actor tail
// samples the action at the very end of its block):
//   env_step_prefetch    every thread requests its seven words of the three planes that survive the shift (they do not depend on anything)
//   env_step_candidates  three threads: the action acts on the env only through the paddle direction (action % n_actions) % 3, so the three
//                        possible transitions (state, reward, done, ...) are computed up front                      [needs a barrier after]
//   env_step_early       the shifted stack is stored, and the NEW plane is the previous newest plane with the ball moved: only the words under
//                        the old and the new 2x2 ball are repainted (env_word); scenery, walls and top bar never change, the paddle rows are
//                        left to the finish, and a brick goes only on a reward (2 % of the steps).  After a reset: four copies of a full repaint
//   env_step_finish      after the action: pick the candidate, publish state / reward / done, paint the 42 words of the paddle rows, repaint
//                        the brick rows if a brick went.  Same bytes as painting every pixel afterwards (tests/test_env.py, test_gpu_e2e.py)
struct EnvStepArgs {
  uint32_t seed; int32_t max_steps; cbm_env_state* st; const uint8_t* obs_prev; uint8_t* obs_next; float* reward; uint8_t* done_next;
  uint8_t* firststep_next;   // obs_next == nullptr: no env step
};
struct EnvCand { cbm_env_state s; EnvOut out; };
struct EnvShared { EnvCand cand[3]; cbm_env_state pre; EnvGame gm; int32_t reset, old_ball_x, old_ball_y; };   // lives in LDS: the brick words are indexed dynamically
#if defined(__HIPCC__)
#define ENV_PLANE_WORDS 1764
// (a native vector type: HIP's uint4 is a struct whose copies between memory spaces are memcpys through a private array the compiler then
// moves to LDS — see env_step_candidates)
typedef unsigned int env_u32x4 __attribute__((ext_vector_type(4)));
#define ENV_PLANE_Q 441        // 16-byte pieces of a plane (a frame stack is 4 x 441 of them, 16-byte aligned in the ring)
// NT threads work on the planes; thread t owns pieces t, t + NT, ... of every plane (NJ = ceil(441 / NT) of them)
template <int NT> struct EnvPieces { static constexpr int NJ = (ENV_PLANE_Q + NT - 1) / NT; env_u32x4 older[NJ][3], nw[NJ]; };
template <int NT>
static __device__ __forceinline__ void env_step_prefetch(const EnvStepArgs& a, int e, int t, EnvPieces<NT>& pc) {
  const env_u32x4* p = reinterpret_cast<const env_u32x4*>(a.obs_prev + (size_t)e * CBM_FRAME);
#pragma unroll
  for (int j = 0; j < EnvPieces<NT>::NJ; ++j) {
    const int i = min(t + NT * j, ENV_PLANE_Q - 1);
    pc.older[j][0] = p[ENV_PLANE_Q + i]; pc.older[j][1] = p[2 * ENV_PLANE_Q + i]; pc.older[j][2] = p[3 * ENV_PLANE_Q + i];
  }
}
static __device__ __forceinline__ void env_q_set(env_u32x4& v, int c, uint32_t x) {
  v.x = c == 0 ? x : v.x; v.y = c == 1 ? x : v.y; v.z = c == 2 ? x : v.z; v.w = c == 3 ? x : v.w;
}
static __device__ __forceinline__ bool env_paddle_word(int i) { return i >= 78 * 21 && i < 80 * 21; }
// a word of the two paddle rows (y = 78, 79): nothing but the paddle and the side walls can be there (the ball stays above row 77, the scenery
// above row 76) — env_pixel's bytes without its case analysis
static __device__ __forceinline__ uint32_t env_paddle_row_word(int paddle_x, int paddle_w, int i) {
  const int y = i / 21, x = 4 * (i - 21 * y);
  uint32_t w = 0u;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int xx = x + p;
    const uint32_t v = (xx >= paddle_x && xx < paddle_x + paddle_w) ? 200u : ((xx == 0 || xx == 83) ? 142u : 0u);
    w |= v << (8 * p);
  }
  return w;
}
// The state travels as 16 plain words (sw = the words of a.st[e], loaded by the caller: the actor tail requests them before its other loads so that
// this arithmetic runs under their latency).  No struct lives in a thread's private memory here — the brick words are indexed dynamically, and a
// private array with dynamic indices is moved to LDS by the compiler, which then reads the dispatch packet for the block shape at kernel entry
// (a host-memory round trip: 9 us per launch when it was measured)
#define ENV_STATE_WORDS 16
static_assert(sizeof(cbm_env_state) == 4 * ENV_STATE_WORDS, "cbm_env_state is 16 words");
static __device__ __forceinline__ void env_state_load_words(const cbm_env_state* st, int e, uint32_t (&sw)[ENV_STATE_WORDS]) {
  const env_u32x4* p = reinterpret_cast<const env_u32x4*>(st + e);
#pragma unroll
  for (int q = 0; q < 4; ++q) { const env_u32x4 v = p[q]; sw[4 * q] = v.x; sw[4 * q + 1] = v.y; sw[4 * q + 2] = v.z; sw[4 * q + 3] = v.w; }
}
static __device__ __forceinline__ void env_step_candidates(const EnvStepArgs& a, int e, EnvShared& sh, int t, const uint32_t (&sw)[ENV_STATE_WORDS]) {
  cbm_env_state* c = &sh.cand[t].s;
  uint32_t* cw = reinterpret_cast<uint32_t*>(c);
#pragma unroll
  for (int i = 0; i < ENV_STATE_WORDS; ++i) cw[i] = sw[i];
  const int was_reset = c->needs_reset, obx = c->ball_x, oby = c->ball_y;
  const uint32_t b0 = c->bricks[0], b1 = c->bricks[1], b2 = c->bricks[2];
  if (t == 0) sh.gm = env_game(c->game);
  sh.cand[t].out = env_transition(c, a.seed, (uint32_t)e, t, a.max_steps);   // action = t: every preset has >= 4 actions, so its direction is t
  if (t == 0) {
    sh.reset = was_reset;
    sh.old_ball_x = obx; sh.old_ball_y = oby;
    // what the early paint shows: candidate 0's ball / episode (the same in all three), and the bricks as they stand BEFORE the step — a reset
    // starts from the new episode's full wall, otherwise nothing has gone yet
    uint32_t* pw = reinterpret_cast<uint32_t*>(&sh.pre);
#pragma unroll
    for (int i = 0; i < ENV_STATE_WORDS; ++i) pw[i] = cw[i];
    if (!was_reset) { sh.pre.bricks[0] = b0; sh.pre.bricks[1] = b1; sh.pre.bricks[2] = b2; }
  }
}
// The new plane's pieces stay in registers (pc.nw) for the finish: a piece is re-stored by the thread that stored it first (program order, no fence).
template <int NT>
static __device__ __forceinline__ void env_step_early(const EnvStepArgs& a, int e, const EnvShared& sh, int t, EnvPieces<NT>& pc) {
  constexpr int NJ = EnvPieces<NT>::NJ;
  const cbm_env_state* pre = &sh.pre;
  const EnvGame& gm = sh.gm;
#pragma unroll
  for (int j = 0; j < NJ; ++j) pc.nw[j] = pc.older[j][2];      // the new plane starts as the previous newest one ...
  if (sh.reset) {   // new episode (about one step in 800): the stack is four copies of the first frame; its paddle rows come from the finish
    uint32_t* o32 = reinterpret_cast<uint32_t*>(a.obs_next + (size_t)e * CBM_FRAME);
#pragma unroll 1
    for (int i = t; i < ENV_PLANE_WORDS; i += NT) {
      if (env_paddle_word(i)) continue;
      const uint32_t w = env_word(pre, gm, i);
      o32[i] = w; o32[ENV_PLANE_WORDS + i] = w; o32[2 * ENV_PLANE_WORDS + i] = w; o32[3 * ENV_PLANE_WORDS + i] = w;
    }
    return;
  }
  env_u32x4* o4 = reinterpret_cast<env_u32x4*>(a.obs_next + (size_t)e * CBM_FRAME);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i4 = t + NT * j;
    if (i4 < ENV_PLANE_Q) { o4[i4] = pc.older[j][0]; o4[ENV_PLANE_Q + i4] = pc.older[j][1]; o4[2 * ENV_PLANE_Q + i4] = pc.older[j][2]; }
  }
  // ... with the (at most eight) words under the old and the new 2x2 ball repainted
  const int obx = sh.old_ball_x, oby = sh.old_ball_y, nbx = pre->ball_x, nby = pre->ball_y;
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    const int bx = k < 4 ? obx : nbx, by = k < 4 ? oby : nby;
    const int r = by + ((k >> 1) & 1), wx = (bx >> 2) + (k & 1), i = r * 21 + wx, i4 = i >> 2;
    if (r < 84 && wx < 21 && wx <= ((bx + 1) >> 2) && !env_paddle_word(i)) {
      bool mine = false;
#pragma unroll
      for (int j = 0; j < NJ; ++j) mine = mine || i4 == t + NT * j;
      if (mine) {
        const uint32_t w = env_word(pre, gm, i);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (i4 == t + NT * j) env_q_set(pc.nw[j], i & 3, w);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i4 = t + NT * j;
    if (i4 < ENV_PLANE_Q) o4[3 * ENV_PLANE_Q + i4] = pc.nw[j];   // (paddle-row words are still the previous step's: the finish re-stores those pieces)
  }
}
// the paddle direction of an action (the only way it acts on the env): what indexes the three candidates
static __device__ __forceinline__ int env_action_dir(const EnvShared& sh, int32_t action) { return (action % sh.gm.n_actions) % 3; }
template <int NT>
static __device__ __forceinline__ void env_step_finish(const EnvStepArgs& a, int e, const EnvShared& sh, int dir, int t, EnvPieces<NT>& pc) {
  constexpr int NJ = EnvPieces<NT>::NJ;
  const EnvGame& gm = sh.gm;
  const cbm_env_state* s = &sh.cand[dir].s;
  if (t == 0) {
    const EnvOut out = sh.cand[dir].out;
    const env_u32x4* sw4 = reinterpret_cast<const env_u32x4*>(s);
    env_u32x4* dst4 = reinterpret_cast<env_u32x4*>(a.st + e);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst4[q] = sw4[q];
    a.reward[e] = out.reward;
    a.done_next[e] = out.done;
    if (a.firststep_next) a.firststep_next[e] = out.firststep;
  }
  if (sh.reset) {   // the paddle rows of all four copies (nothing else can have changed: a reset step ignores the action)
    uint32_t* o32 = reinterpret_cast<uint32_t*>(a.obs_next + (size_t)e * CBM_FRAME);
    if (t < 42) {
      const int i = 78 * 21 + t;
      const uint32_t w = env_paddle_row_word(s->paddle_x, gm.paddle_w, i);
      o32[i] = w; o32[ENV_PLANE_WORDS + i] = w; o32[2 * ENV_PLANE_WORDS + i] = w; o32[3 * ENV_PLANE_WORDS + i] = w;
    }
    return;
  }
  const bool bricks = s->bricks[0] != sh.pre.bricks[0] || s->bricks[1] != sh.pre.bricks[1] || s->bricks[2] != sh.pre.bricks[2];   // block-uniform
  env_u32x4* o4 = reinterpret_cast<env_u32x4*>(a.obs_next + (size_t)e * CBM_FRAME);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i4 = t + NT * j, lo = 4 * i4;
    if (i4 >= ENV_PLANE_Q) continue;
    bool hit = false;
    if (lo + 3 >= 78 * 21 && lo < 80 * 21) {                       // pieces 409 .. 419
      hit = true;
      const int px = s->paddle_x, pw = gm.paddle_w;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (env_paddle_word(lo + c)) env_q_set(pc.nw[j], c, env_paddle_row_word(px, pw, lo + c));
    }
    if (bricks && lo + 3 >= 17 * 21 && lo < 35 * 21) {            // a reward took a brick (2 % of the steps): the brick rows again
      hit = true;
#pragma unroll 1
      for (int c = 0; c < 4; ++c)
        if (lo + c >= 17 * 21 && lo + c < 35 * 21) env_q_set(pc.nw[j], c, env_word(s, gm, lo + c));
    }
    if (hit) o4[3 * ENV_PLANE_Q + i4] = pc.nw[j];
  }
}
#endif
