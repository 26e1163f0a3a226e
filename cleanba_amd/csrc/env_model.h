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


// ---- one env's step by one 256-thread block, in two parts so that a caller can put work between them:
//   env_step_prefetch  the three older frame planes of the stack (they do not depend on the action) -> registers
//   env_step_block     transition (thread 0), then every thread paints its words of the new plane and writes the shifted stack
struct EnvStepArgs {
  uint32_t seed; int32_t max_steps; cbm_env_state* st; const uint8_t* obs_prev; uint8_t* obs_next; float* reward; uint8_t* done_next;
  uint8_t* firststep_next;   // obs_next == nullptr: no env step
};
#if defined(__HIPCC__)
static __device__ __forceinline__ void env_step_prefetch(const EnvStepArgs& a, int e, uint32_t (&older)[7][3]) {
  const uint32_t* p32 = reinterpret_cast<const uint32_t*>(a.obs_prev + (size_t)e * CBM_FRAME);   // 1764 words per plane
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int i = min((int)threadIdx.x + 256 * j, 1763);
    older[j][0] = p32[1764 + i]; older[j][1] = p32[2 * 1764 + i]; older[j][2] = p32[3 * 1764 + i];
  }
}
static __device__ __forceinline__ void env_step_block(const EnvStepArgs& a, int e, int32_t action, const uint32_t (&older)[7][3]) {
  __shared__ cbm_env_state s;
  __shared__ EnvOut out;
  __shared__ EnvGame gm;
  if (threadIdx.x == 0) {
    s = a.st[e];
    gm = env_game(s.game);
    out = env_transition(&s, a.seed, (uint32_t)e, action, a.max_steps);
    a.st[e] = s;
    a.reward[e] = out.reward;
    a.done_next[e] = out.done;
    if (a.firststep_next) a.firststep_next[e] = out.firststep;
  }
  __syncthreads();
  uint32_t* o32 = reinterpret_cast<uint32_t*>(a.obs_next + (size_t)e * CBM_FRAME);
  const bool rs = out.was_reset;
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
#endif
