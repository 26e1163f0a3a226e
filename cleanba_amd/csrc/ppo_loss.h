// ppo_loss.h — the PPO loss head of one sample on its 32 lanes (ppo:516-577), shared by ppo_loss_kernel (pointwise.hip) and the fused
// heads + loss + heads-dgrad kernel (gemm_layers.hip).  Lane j holds the logit of action j (lanes >= A idle); the exponentials run in parallel,
// the softmax sums are taken in ascending action order by a shuffle walk (the same order, hence the same bits, as a serial loop and as the
// oracle).  Returns this lane's dL/d(logit j) (lane A: dL/dvalue, other lanes 0); lane 0's `st` holds the sample's four statistics.
#pragma once
#include "cbm_internal.h"
#include <float.h>

struct PpoSampleStats { float pg, dv2, ent, kl; };
static __device__ __forceinline__ float ppo_loss_lane(float zj, int j, int A, int a, float value, float old_lp, float ad, float tgt, float clip_coef,
                                                      float ent_coef, float vf_coef, float invN, PpoSampleStats& st) {
  const bool act = j < A;
  float mx = act ? zj : -INFINITY;
  for (int o = 16; o > 0; o >>= 1) { const float t = __shfl_xor(mx, o, 32); mx = t > mx ? t : mx; }
  const float ej = act ? cbm_expf(zj - mx) : 0.0f;
  float se = 0.0f;
  for (int q = 0; q < A; ++q) se += __shfl(ej, q, 32);
  const float lse_shift = cbm_logf(se);
  const float za = __shfl(zj, a, 32);
  const float newlp = (za - mx) - lse_shift;
  const float lse = lse_shift + mx;
  float zn = zj - lse;
  if (zn < -FLT_MAX) zn = -FLT_MAX;
  float mx2 = act ? zn : -INFINITY;
  for (int o = 16; o > 0; o >>= 1) { const float t = __shfl_xor(mx2, o, 32); mx2 = t > mx2 ? t : mx2; }
  const float e2 = act ? cbm_expf(zn - mx2) : 0.0f;
  float s2 = 0.0f;
  for (int q = 0; q < A; ++q) s2 += __shfl(e2, q, 32);
  const float pj = e2 / s2;
  const float tj = act ? zn * pj : 0.0f;
  float ent = 0.0f;
  for (int q = 0; q < A; ++q) ent += __shfl(tj, q, 32);
  ent = -ent;
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
