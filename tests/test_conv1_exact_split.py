"""The arithmetic claim behind the exact-product conv1 kernels (csrc/conv1.hip: c1_split3, c1_px8_bf16), checked on the CPU in numpy:

  * an fp32 value v is cut into t1 + t2 + t3 by keeping the upper 16 bits of v, of v - t1 and of v - t1 - t2 (truncation): every term is
    bf16-representable (its low 16 bits are zero), the two subtractions are exact, and t1 + t2 + t3 == v EXACTLY — including subnormal
    negative values and values whose mantissa is all ones.  Domain: |v| >= 2^-103 (or 0), so that the third term is a NORMAL number; below that the
    remainder is subnormal, its low 16 bits need not be zero, and the kernel's third term (upper 16 bits) is off by less than 2^-133 — weights / 255
    and loss gradients live thirty orders of magnitude above that (second test);
  * a pixel 0..255 is exact in bf16 (the upper 16 bits of its fp32 encoding lose nothing);
  * every product pixel x term has at most 16 significant bits, so it is exact in fp32 — what v_mfma_f32_32x32x16_bf16 sums is therefore the exact
    products; the kernel's result differs from the oracle's fmaf chain only by where the fp32 sums round.
No GPU involved: this pins the restatement, tests/test_gpu_conv1_exact.py holds the kernels to the oracle."""
import numpy as np


def split3(v):
    v = np.asarray(v, np.float32)
    t1 = (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    r1 = (v - t1).astype(np.float32)
    t2 = (r1.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    r2 = (r1 - t2).astype(np.float32)
    return t1, t2, r2, r1


def _values():
    rng = np.random.default_rng(7)
    parts = [rng.normal(0, 0.05, 200000), rng.normal(0, 1e-4, 100000) / 255.0, rng.uniform(-2, 2, 100000),
             np.array([0.0, -0.0, 1.0, -1.0, 255.0, 1.0 / 255.0, np.float32(1.9999999), np.float32(-3.4e38), np.float32(2.0 ** -100), np.float32(1e-30),
                       np.nextafter(np.float32(1), np.float32(2)), np.nextafter(np.float32(1), np.float32(0))], np.float64)]
    return np.concatenate(parts).astype(np.float32)


def test_three_truncated_terms_sum_to_the_value_exactly_and_are_bf16():
    v = _values()
    t1, t2, t3, r1 = split3(v)
    for t in (t1, t2, t3):
        assert (t.view(np.uint32) & np.uint32(0xFFFF) == 0).all(), "a term has bits below bf16's mantissa"
    # the subtractions are exact (checked in float64, which holds any fp32 difference of these magnitudes exactly)
    assert (r1.astype(np.float64) == v.astype(np.float64) - t1.astype(np.float64)).all()
    assert (t3.astype(np.float64) == r1.astype(np.float64) - t2.astype(np.float64)).all()
    assert (t1.astype(np.float64) + t2.astype(np.float64) + t3.astype(np.float64) == v.astype(np.float64)).all()
    # truncation: every term carries the sign of v (or is zero), magnitudes fall by at least 2^-7 per term
    nz = v != 0
    assert (np.sign(t1[nz]) == np.sign(v[nz])).all()
    assert (np.abs(t2) <= np.abs(t1) * 2.0 ** -7).all() and (np.abs(t3) <= np.abs(t2) * 2.0 ** -7 + 0).all()


def test_below_the_domain_the_identity_still_holds_and_the_truncated_third_term_is_off_by_less_than_2_to_minus_133():
    v = np.array([1.2e-38, -3e-38, 2.0 ** -110, 5e-36, -7.7e-33], np.float32)
    t1, t2, t3, _ = split3(v)
    assert (t1.astype(np.float64) + t2.astype(np.float64) + t3.astype(np.float64) == v.astype(np.float64)).all()
    t3_bf16 = (t3.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)          # what the kernel packs
    assert (np.abs(t3.astype(np.float64) - t3_bf16.astype(np.float64)) < 2.0 ** -133).all()


def test_pixels_are_exact_in_bf16_and_products_are_exact_in_fp32():
    px = np.arange(256, dtype=np.float32)
    assert (px.view(np.uint32) & np.uint32(0xFFFF) == 0).all(), "an integer 0..255 needs more than bf16's 8 significant bits?"
    v = _values()[:20000]
    v = v[np.abs(v) < 1e30]        # (keep px * v finite)
    v = v[(v == 0) | (np.abs(v) > 1e-25)]   # fp32-normal remainders: conv weights / 255 and dY live far above the subnormal range
    for t in split3(v)[:3]:
        p64 = px[:, None].astype(np.float64) * t[None, :].astype(np.float64)          # exact in float64 (8 x 8 significant bits)
        p32 = (px[:, None] * t[None, :]).astype(np.float32)                            # what an fp32 multiply (or the MFMA's product) yields
        assert (p32.astype(np.float64) == p64).all()
    # and the sum of the three exact products is the exact product px * v (as real numbers): the split loses nothing
    t1, t2, t3, _ = split3(v)
    exact = px[:, None].astype(np.float64) * v[None, :].astype(np.float64)
    s = px[:, None].astype(np.float64) * t1[None, :] + px[:, None].astype(np.float64) * t2[None, :] + px[:, None].astype(np.float64) * t3[None, :]
    assert (s == exact).all()
