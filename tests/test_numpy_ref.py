"""The C oracle against tests/numpy_ref.py — an independently written float64 numpy restatement (definitions as sums, gradients by finite
differences) — and cbm_logf / cbm_expf over the EXACT argument sets the action sampling feeds them."""
import math

import numpy as np

import numpy_ref as R

A = 18


def test_gae_matches_the_forward_sum_definition(oracle):
    rng = np.random.default_rng(11)
    T, B = 24, 5
    r = (rng.random((T, B)) < 0.2).astype(np.float32)
    v = rng.normal(size=(T, B)).astype(np.float32)
    d = (rng.random((T, B)) < 0.1).astype(np.uint8)
    nv = rng.normal(size=B).astype(np.float32)
    nd = (rng.random(B) < 0.2).astype(np.uint8)
    adv, tgt = oracle.gae(r, v, d, nv, nd)
    adv_r, tgt_r = R.gae(r, v, d, nv, nd)
    np.testing.assert_allclose(adv, adv_r, rtol=0, atol=2e-5)
    np.testing.assert_allclose(tgt, tgt_r, rtol=0, atol=2e-5)


def test_vtrace_matches_the_sum_definition(oracle):
    rng = np.random.default_rng(12)
    T, B = 20, 6
    V = rng.normal(size=(T + 1, B)).astype(np.float32)
    r = (rng.random((T, B)) < 0.3).astype(np.float32)
    disc = (0.99 * (rng.random((T, B)) > 0.1)).astype(np.float32)
    rho = np.exp(rng.normal(0, 0.5, size=(T, B))).astype(np.float32)
    err, pg, q = oracle.vtrace(V[:-1], V[1:], r, disc, rho)
    err_r, pg_r, q_r = R.vtrace(V[:-1], V[1:], r, disc, rho)
    np.testing.assert_allclose(err, err_r, rtol=0, atol=2e-5)
    np.testing.assert_allclose(pg, pg_r, rtol=0, atol=2e-5)
    np.testing.assert_allclose(q, q_r, rtol=0, atol=2e-5)


def test_ppo_loss_head_values_and_finite_difference_gradients(oracle):
    rng = np.random.default_rng(13)
    N = 10
    lg = rng.normal(size=(N, A)).astype(np.float32)
    val = rng.normal(size=N).astype(np.float32)
    act = rng.integers(0, A, N).astype(np.int32)
    olp = (-np.log(A) + 0.3 * rng.normal(size=N)).astype(np.float32)
    ad = rng.normal(size=N).astype(np.float32)
    tg = rng.normal(size=N).astype(np.float32)
    st, dl, dv = oracle.ppo_loss_head(lg, val, act, olp, ad, tg)
    ref = R.ppo_loss(lg, val, act, olp, ad, tg)
    np.testing.assert_allclose(st, ref, rtol=1e-5, atol=1e-6)
    gl = R.fd_grad(lambda z: R.ppo_loss(z, val, act, olp, ad, tg)[0], lg)
    gv = R.fd_grad(lambda z: R.ppo_loss(lg, z, act, olp, ad, tg)[0], val)
    np.testing.assert_allclose(dl, gl, rtol=0, atol=2e-6)
    np.testing.assert_allclose(dv, gv, rtol=0, atol=2e-6)


def test_impala_loss_head_values_and_finite_difference_gradients(oracle):
    rng = np.random.default_rng(14)
    T1, B = 6, 3
    lg = rng.normal(size=(T1, B, A)).astype(np.float32)
    mu = rng.normal(size=(T1, B, A)).astype(np.float32)
    val = rng.normal(size=(T1, B)).astype(np.float32)
    act = rng.integers(0, A, (T1, B)).astype(np.int32)
    rew = (rng.random((T1, B)) < 0.4).astype(np.float32)
    dn = (rng.random((T1, B)) < 0.2).astype(np.uint8)
    fs = (rng.random((T1, B)) < 0.2).astype(np.uint8)
    st, dl, dv = oracle.impala_loss_head(lg, val, mu, act, rew, dn, fs)
    lpa, lp, err, pg_adv, mask = R.impala_loss(lg, val, mu, act, rew, dn, fs)
    consts = (err + val.astype(np.float64)[:-1], pg_adv, mask, act[:-1])
    ref = R.impala_loss_value(lg, val, consts)
    np.testing.assert_allclose(st, ref, rtol=1e-5, atol=1e-5)
    gl = R.fd_grad(lambda z: R.impala_loss_value(z, val, consts)[0], lg)
    gv = R.fd_grad(lambda z: R.impala_loss_value(lg, z, consts)[0], val)
    np.testing.assert_allclose(dl, gl, rtol=0, atol=5e-6)
    np.testing.assert_allclose(dv, gv, rtol=0, atol=5e-6)
    assert np.abs(dl[-1]).max() == 0 and np.abs(dv[-1]).max() == 0     # the bootstrap row only feeds constants


def test_optimizers_match_the_optax_definitions(oracle):
    rng = np.random.default_rng(15)
    n = 4096
    for scale in (1e-4, 3.0):     # below / above the clipping threshold
        p = rng.normal(size=n).astype(np.float32)
        m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
        pr, mr, vr = p.astype(np.float64), np.zeros(n), np.zeros(n)
        for step in range(3):
            g = (scale * rng.normal(size=n)).astype(np.float32)
            oracle.adam_step(p, g, m, v, 0.5, 2.5e-4, count=step + 1)
            pr, mr, vr = R.adam(pr, g, mr, vr, step, 2.5e-4)
            np.testing.assert_allclose(p, pr, rtol=3e-7, atol=1e-7)      # one float32 ulp of the parameter
        q = rng.normal(size=n).astype(np.float32)
        nu = np.zeros(n, np.float32)
        qr, nur = q.astype(np.float64), np.zeros(n)
        for step in range(3):
            g = (scale * 30 * rng.normal(size=n)).astype(np.float32)
            oracle.rmsprop_step(q, g, nu, 40.0, 6e-4)
            qr, nur = R.rmsprop(qr, g, nur, 6e-4)
            np.testing.assert_allclose(q, qr, rtol=3e-7, atol=2e-6)
    # clip_by_global_norm's strict '<': a gradient of norm exactly max_norm takes the scaling branch, which is the identity there
    g = np.zeros(16, np.float64)
    g[0] = 0.5
    assert np.array_equal(R.clip_by_global_norm(g, 0.5), g)
    gs = [rng.normal(size=8) for _ in range(4)]
    np.testing.assert_allclose(R.multisteps_mean(gs), np.mean(gs, axis=0), rtol=1e-12)


def test_logf_on_every_value_the_sampler_can_feed_it(oracle):
    """jax.random.uniform(float32) takes exactly the 2^23 values k / 2^23; the Gumbel perturbation is log(-log(u)) (ppo:258-259).  Both
    logs over the WHOLE domain against the correctly rounded float64 result: cbm_logf may be off by at most one float32 ulp anywhere, and
    wherever it is exact the perturbation is the float every correctly rounded libm (XLA's included) produces."""
    k = np.arange(1 << 23, dtype=np.uint32)
    u = oracle.bits_to_uniform_v(k << np.uint32(9))
    assert u[0] == 0.0 and u[-1] == np.float32(1.0 - 2.0 ** -23) and np.array_equal(u, (k.astype(np.float64) / 2 ** 23).astype(np.float32))
    inner = oracle.logf_v(u)                                   # log(u) in [-15.94, 0), log(0) = -inf
    with np.errstate(divide="ignore"):
        ref_inner = np.log(u.astype(np.float64))
    assert inner[0] == -np.inf
    cr = ref_inner.astype(np.float32)                          # correctly rounded
    ulp = np.abs(np.spacing(cr[1:]))
    assert np.abs(inner[1:].astype(np.float64) - ref_inner[1:]).max() <= 1.0 * ulp.max()
    off_inner = np.abs(inner[1:] - cr[1:]) > 0
    assert (np.abs(inner[1:] - cr[1:]) <= ulp).all()
    x = -inner[1:]
    outer = oracle.logf_v(x)
    ref_outer = np.log(x.astype(np.float64))
    cro = ref_outer.astype(np.float32)
    ulpo = np.abs(np.spacing(np.where(cro == 0, np.float32(1e-7), cro)))
    assert (np.abs(outer - cro) <= ulpo).all()
    off_outer = np.abs(outer - cro) > 0
    # the fraction of the domain where cbm_logf is not the correctly rounded float (reported, and bounded so that a regression shows)
    fi, fo = off_inner.mean(), off_outer.mean()
    print(f"cbm_logf vs correctly rounded log over the sampler's domain: inner {fi:.4%} off by one ulp, outer {fo:.4%}")
    assert fi < 0.12 and fo < 0.12


def test_expf_on_the_softmax_domain(oracle):
    """log_softmax feeds exp() with shifted logits in (-inf, 0]: a dense sweep of that range.  cbm_expf is within 1.01 float32 ulp of the
    true value wherever the result is a normal float, and flushes to 0 where it would be subnormal (x < -87.33), which a softmax sum >= 1
    cannot see."""
    x = -np.concatenate([np.linspace(0, 30, 2_000_001), 2.0 ** np.linspace(-30, 6.7, 200_001), [87.3, 88.0, 100.0, 104.0]]).astype(np.float32)
    got = oracle.expf_v(x)
    ref = np.exp(x.astype(np.float64))
    normal = ref >= 1.1754944e-38
    ulp = np.abs(np.spacing(ref.astype(np.float32)))
    err = np.abs(got.astype(np.float64) - ref)[normal] / ulp[normal]
    assert err.max() <= 1.01, err.max()
    assert (got[~normal] == 0.0).all()
    assert got[0] == 1.0
    print(f"cbm_expf over the softmax domain: max error {err.max():.3f} ulp, {np.mean(err > 0.5):.2%} of the points not correctly rounded")
