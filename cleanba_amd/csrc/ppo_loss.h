// ppo_loss.h — the PPO loss head of one sample on its 32 lanes (ppo:516-577), shared by ppo_loss_kernel (pointwise.hip) and the fused
// heads + loss + heads-dgrad kernel (gemm_layers.hip).  Lane j holds the logit of action j (lanes >= A idle); the exponentials run in parallel,
// the softmax sums are taken in ascending action order (the same order, hence the same bits, as a serial loop and as the oracle): every lane
// fetches all A terms by shuffles that are ALL requested before the first add — as a walk of A dependent shuffle + add pairs the three sums of a
// sample cost 54 LDS-crossbar round trips (8.8 us of the fused kernel's 23: tools/heads_trace.py).  Returns this lane's dL/d(logit j) (lane A: dL/dvalue, other lanes 0); lane 0's `st` holds the sample's four statistics.
#pragma once
#include "cbm_internal.h"
#include <float.h>

struct PpoSampleStats { float pg, dv2, ent, kl; };
// sum_{q < A} x_q in ascending q, x_q = lane q's value (A <= 28)
static __device__ __forceinline__ float ppo_ordered_sum32(float x, int A) {
  float t[28];
#pragma unroll
  for (int q = 0; q < 28; ++q) t[q] = __shfl(x, q, 32);
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < 28; ++q)
    if (q < A) s += t[q];
  return s;
}
static __device__ __forceinline__ float ppo_loss_lane(float zj, int j, int A, int a, float value, float old_lp, float ad, float tgt, float clip_coef,
                                                      float ent_coef, float vf_coef, float invN, PpoSampleStats& st) {
  const bool act = j < A;
  float mx = act ? zj : -INFINITY;
  for (int o = 16; o > 0; o >>= 1) { const float t = __shfl_xor(mx, o, 32); mx = t > mx ? t : mx; }
  const float ej = act ? cbm_expf(zj - mx) : 0.0f;
  const float se = ppo_ordered_sum32(ej, A);
  const float lse_shift = cbm_logf(se);
  const float za = __shfl(zj, a, 32);
  const float newlp = (za - mx) - lse_shift;
  const float lse = lse_shift + mx;
  float zn = zj - lse;
  if (zn < -FLT_MAX) zn = -FLT_MAX;
  float mx2 = act ? zn : -INFINITY;
  for (int o = 16; o > 0; o >>= 1) { const float t = __shfl_xor(mx2, o, 32); mx2 = t > mx2 ? t : mx2; }
  const float e2 = act ? cbm_expf(zn - mx2) : 0.0f;
  const float s2 = ppo_ordered_sum32(e2, A);
  const float pj = e2 / s2;
  const float tj = act ? zn * pj : 0.0f;
  const float ent = -ppo_ordered_sum32(tj, A);
  const float logratio = newlp - old_lp;
  const float ratio = cbm_expf(logratio);
  const float lo = 1.0f - clip_coef, hi = 1.0f + clip_coef;
  const float rc = ratio < lo ? lo : (ratio > hi ? hi : ratio);
  const float pg1 = -ad * ratio, pg2 = -ad * rc;
  const float pg = pg1 > pg2 ? pg1 : pg2;
  const float dv = value - tgt;
  const float w1 = pg1 > pg2 ? 1.0f : (pg1 == pg2 ? 0.5f : 0.0f);
  const float dclip = (ratio > lo && ratio < hi) ? 1.0f : ((ratio == lo || ratio == hi) ? 0.5f : 0.0f);
  const float dpg_dratio = w1 * (-ad) + (1.0f - w1) * (-ad) * dclip;
  const float c_lp = dpg_dratio * ratio * invN;
  float d = 0.0f;
  if (act) d = c_lp * ((j == a ? 1.0f : 0.0f) - pj) + ent_coef * invN * pj * (zn + ent);
  else if (j == A) d = vf_coef * dv * invN;
  st.pg = pg; st.dv2 = dv * dv; st.ent = ent; st.kl = (ratio - 1.0f) - logratio;
  return d;
}
