/* cbm_math.h — shared scalar numerics for the cleanba-mi hot path.
 *
 * One definition, compiled twice: by hipcc into the gfx950 kernels and by gcc into the
 * CPU oracle (oracle/ includes this header; the product never includes anything from
 * oracle/).  Everything here is written with explicit IEEE operations (no reliance on
 * libm / ocml, no contraction: both sides are compiled with -ffp-contract=off and use
 * fmaf only where it is spelled out), so CPU and GPU produce the same bits.  That is
 * what makes "sampled action indices bit-exact" (BASELINE.json north_star) testable:
 * the Gumbel perturbation log(-log u) is the same float on both sides.
 *
 * What it restates (reference = /root/reference, jax 0.4.8 semantics):
 *   - threefry2x32 / PRNGKey / split / random_bits / uniform  (jax.random, used at
 *     cleanba_ppo.py:256-257, 468-469, 599, 606)
 *   - x / 255.0 on uint8 pixels                                (cleanba_ppo.py:181)
 *   - logf / expf used by log_softmax, logsumexp, softmax, exp(logratio)
 *     (cleanba_ppo.py:258-259, 524-527, 565)
 */
#ifndef CBM_MATH_H
#define CBM_MATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define CBM_HD __host__ __device__ __forceinline__
#else
#define CBM_HD static inline
#endif

/* ------------------------------------------------------------------ bit casts */
CBM_HD float cbm_u2f(uint32_t u) {
  union { uint32_t u; float f; } c; c.u = u; return c.f;
}
CBM_HD uint32_t cbm_f2u(float f) {
  union { uint32_t u; float f; } c; c.f = f; return c.u;
}

/* ------------------------------------------------------------------ threefry2x32
 * Random123 Threefry-2x32, 20 rounds, as used by jax._src.prng.threefry2x32.
 * KAT (tests/test_prng.py): key=(0,0),ctr=(0,0) -> (0x6b200159, 0x99ba4efe). */
CBM_HD uint32_t cbm_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

#define CBM_TF_ROUND(r) do { x0 += x1; x1 = cbm_rotl32(x1, (r)); x1 ^= x0; } while (0)

CBM_HD void cbm_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                             uint32_t* o0, uint32_t* o1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + k0, x1 = c1 + k1;
  CBM_TF_ROUND(13); CBM_TF_ROUND(15); CBM_TF_ROUND(26); CBM_TF_ROUND(6);
  x0 += k1; x1 += k2 + 1u;
  CBM_TF_ROUND(17); CBM_TF_ROUND(29); CBM_TF_ROUND(16); CBM_TF_ROUND(24);
  x0 += k2; x1 += k0 + 2u;
  CBM_TF_ROUND(13); CBM_TF_ROUND(15); CBM_TF_ROUND(26); CBM_TF_ROUND(6);
  x0 += k0; x1 += k1 + 3u;
  CBM_TF_ROUND(17); CBM_TF_ROUND(29); CBM_TF_ROUND(16); CBM_TF_ROUND(24);
  x0 += k1; x1 += k2 + 4u;
  CBM_TF_ROUND(13); CBM_TF_ROUND(15); CBM_TF_ROUND(26); CBM_TF_ROUND(6);
  x0 += k2; x1 += k0 + 5u;
  *o0 = x0; *o1 = x1;
}

/* random_bits(key, 32, [n])[i]  (jax _threefry_random_bits + threefry_2x32's
 * "split the iota in two halves, pad one zero when odd" scheme).  Element i is
 * independent of the others, so a GPU thread can compute just its own. */
CBM_HD uint32_t cbm_random_bits_at(uint32_t k0, uint32_t k1, uint32_t n, uint32_t i) {
  const uint32_t half = (n + 1u) >> 1;          /* size of each half after padding */
  const uint32_t j = (i < half) ? i : i - half; /* lane inside the half            */
  uint32_t c1 = j + half;                       /* second-half counter             */
  if (c1 >= n) c1 = 0u;                         /* the padded element (odd n)      */
  uint32_t o0, o1;
  cbm_threefry2x32(k0, k1, j, c1, &o0, &o1);
  return (i < half) ? o0 : o1;
}

/* split(key, num)[r] = (bits[2r], bits[2r+1]) with bits = random_bits over iota(2*num). */
CBM_HD void cbm_split_at(uint32_t k0, uint32_t k1, uint32_t num, uint32_t r,
                         uint32_t* o0, uint32_t* o1) {
  *o0 = cbm_random_bits_at(k0, k1, 2u * num, 2u * r);
  *o1 = cbm_random_bits_at(k0, k1, 2u * num, 2u * r + 1u);
}

/* uniform [0,1): bitcast((bits >> 9) | 0x3F800000) - 1.0, then max(0, .) */
CBM_HD float cbm_bits_to_uniform(uint32_t bits) {
  float f = cbm_u2f((bits >> 9) | 0x3F800000u) - 1.0f;
  return f > 0.0f ? f : 0.0f;
}

/* ------------------------------------------------------------------ pixel scale
 * x / 255.0f, correctly rounded, without a hardware divide: 1/255 as a two-term float
 * (hi = fl(1/255), lo = fl(1/255 - hi)); x*hi is exact inside the fma, so the result is
 * fl(x*hi + fl(x*lo)) = fl(x/255) for every byte (3 instructions per pixel with the
 * int->float conversion; the earlier multiply + Newton residual took 4).
 * tests/test_oracle_prng.py checks all 256 inputs against true division. */
CBM_HD float cbm_u8_unit(uint32_t x) {
  const float hi = 0x1.010102p-8f;    /* fl(1/255) */
  const float lo = -0x1.fdfdfep-33f;  /* fl(1/255 - hi) */
  const float xf = (float)x;
  return fmaf(xf, hi, xf * lo);
}

/* ------------------------------------------------------------------ logf
 * Cephes-style single-precision log with explicit fma Horner steps.
 * log(0) = -inf, log(+inf) = +inf, log(x<0) = NaN.  ~1 ulp on normal inputs. */
CBM_HD float cbm_logf(float x) {
  uint32_t ux = cbm_f2u(x);
  if (ux == 0u || ux == 0x80000000u) return -INFINITY;
  if (ux >> 31) return NAN;
  if (ux >= 0x7F800000u) return x; /* +inf or NaN */
  int e = 0;
  if (ux < 0x00800000u) { /* subnormal: scale by 2^23 */
    x = x * 8388608.0f;
    ux = cbm_f2u(x);
    e = -23;
  }
  e += (int)(ux >> 23) - 126;                       /* x = m * 2^e, m in [0.5,1) */
  float m = cbm_u2f((ux & 0x007FFFFFu) | 0x3F000000u);
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  const float z = m * m;
  float y = 7.0376836292E-2f;
  y = fmaf(y, m, -1.1514610310E-1f);
  y = fmaf(y, m, 1.1676998740E-1f);
  y = fmaf(y, m, -1.2420140846E-1f);
  y = fmaf(y, m, 1.4249322787E-1f);
  y = fmaf(y, m, -1.6668057665E-1f);
  y = fmaf(y, m, 2.0000714765E-1f);
  y = fmaf(y, m, -2.4999993993E-1f);
  y = fmaf(y, m, 3.3333331174E-1f);
  y = y * m * z;
  const float fe = (float)e;
  y = fmaf(-2.12194440e-4f, fe, y);
  y = fmaf(-0.5f, z, y);
  float r = m + y;
  r = fmaf(0.693359375f, fe, r);
  return r;
}

/* ------------------------------------------------------------------ expf
 * Cephes-style expf; returns 0 below -87.3 and +inf above 88.72. */
CBM_HD float cbm_expf(float x) {
  if (x != x) return x;
  if (x > 88.72283905206835f) return INFINITY;
  if (x < -87.33654475055310898657f) return 0.0f;
  float fn = floorf(fmaf(1.44269504088896341f, x, 0.5f));
  float r = fmaf(fn, -0.693359375f, x);
  r = fmaf(fn, 2.12194440e-4f, r);
  const float z = r * r;
  float y = 1.9875691500E-4f;
  y = fmaf(y, r, 1.3981999507E-3f);
  y = fmaf(y, r, 8.3334519073E-3f);
  y = fmaf(y, r, 4.1665795894E-2f);
  y = fmaf(y, r, 1.6666665459E-1f);
  y = fmaf(y, r, 5.0000001201E-1f);
  y = fmaf(y, z, r);
  y = y + 1.0f;
  const int n = (int)fn;                 /* in [-126, 128] given the clamps */
  if (n > 127) {                         /* 2^128 is not representable: two steps */
    return y * cbm_u2f((uint32_t)(127 + 127) << 23) * 2.0f;
  }
  if (n < -126) {
    return y * cbm_u2f((uint32_t)(n + 127 + 24) << 23) * 5.9604644775390625e-8f;
  }
  return y * cbm_u2f((uint32_t)(n + 127) << 23);
}

#endif /* CBM_MATH_H */
