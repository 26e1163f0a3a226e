"""Cross-implementation pin: C oracle (fp32, fixed chain order) vs torch-CPU float64 autograd."""
import numpy as np
import torch

from helpers import make_frames, make_params
import torch_ref as tr

A = 18


def test_forward_matches_torch(oracle):
    P = make_params(A, 1)
    obs = make_frames(6, 2)
    logits, value, acts = oracle.nature_forward(P, A, obs, save_acts=True)
    _, Pt = tr.unpack_nature(P, A)
    lt, vt = tr.nature_forward(Pt, obs)
    np.testing.assert_allclose(logits, lt.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(value, vt.numpy(), rtol=0, atol=2e-6)
    # idx indirection and ksplit change only rounding
    l2, v2 = oracle.nature_forward(P, A, obs, idx=[5, 0, 3], ksplit=14)
    np.testing.assert_allclose(l2, logits[[5, 0, 3]], rtol=0, atol=2e-6)
    a1, a2, a3, hid = oracle.split_acts(acts, 6)
    assert a1.min() >= 0 and (a1 > 0).mean() > 0.05 and (hid > 0).mean() > 0.05


def test_ppo_loss_and_grads_match_torch(oracle):
    rng = np.random.default_rng(3)
    N = 24
    P = make_params(A, 4)
    obs = make_frames(40, 5)
    idx = rng.permutation(40)[:N].astype(np.int32)
    actions = rng.integers(0, A, N).astype(np.int32)
    old_lp = (-np.log(A) + 0.2 * rng.normal(size=N)).astype(np.float32)
    adv = rng.normal(size=N).astype(np.float32)
    tgt = rng.normal(size=N).astype(np.float32)
    stats, grads, logits, value = oracle.ppo_loss_grad(P, A, obs, idx, actions, old_lp, adv, tgt)
    flat, Pt = tr.unpack_nature(P, A, requires_grad=True)
    lt, vt = tr.nature_forward(Pt, obs[idx])
    loss, (pg, v, e, kl) = tr.ppo_loss(lt, vt, actions, old_lp, adv, tgt)
    loss.backward()
    ref = np.array([loss.item(), pg.item(), v.item(), e.item(), kl.item()])
    np.testing.assert_allclose(stats, ref, rtol=1e-5, atol=1e-6)
    g = flat.grad.numpy()
    assert np.abs(g).max() > 1e-4
    np.testing.assert_allclose(grads, g, rtol=0, atol=1e-5 * np.abs(g).max())
    # per-layer relative check so small layers are not hidden by the largest one
    for name, (o, shp) in oracle.nature_layout(A).items():
        n = int(np.prod(shp))
        ga, gb = grads[o:o + n], g[o:o + n]
        assert np.abs(ga - gb).max() <= 2e-5 * max(np.abs(gb).max(), 1e-8), name


def test_ppo_identities(oracle):
    # ratio == 1  =>  approx_kl = 0, pg = -mean(adv); uniform logits => entropy = ln A
    N = 16
    logits = np.zeros((N, A), np.float32)
    value = np.zeros(N, np.float32)
    actions = np.arange(N, dtype=np.int32) % A
    old_lp = np.full(N, -np.log(A), np.float32)
    adv = np.linspace(-1, 1, N).astype(np.float32)
    stats, dlog, dval = oracle.ppo_loss_head(logits, value, actions, old_lp, adv, np.ones(N, np.float32))
    assert abs(stats[4]) < 1e-6 and abs(stats[1] + adv.mean()) < 1e-6
    assert abs(stats[3] - np.log(A)) < 1e-6 and abs(stats[2] - 0.5) < 1e-6


def test_impala_loss_and_grads_match_torch(oracle):
    rng = np.random.default_rng(7)
    T1, Bm = 5, 3
    N = T1 * Bm
    P = make_params(A, 8)
    obs = make_frames(N, 9)
    mu = rng.normal(0, 0.3, size=(T1, Bm, A)).astype(np.float32)
    actions = rng.integers(0, A, (T1, Bm)).astype(np.int32)
    rewards = (rng.random((T1, Bm)) < 0.3).astype(np.float32)
    dones = (rng.random((T1, Bm)) < 0.2).astype(np.uint8)
    first = (rng.random((T1, Bm)) < 0.2).astype(np.uint8)
    stats, grads = oracle.impala_loss_grad(P, A, obs, None, T1, Bm, mu, actions, rewards, dones, first)
    flat, Pt = tr.unpack_nature(P, A, requires_grad=True)
    lt, vt = tr.nature_forward(Pt, obs)
    loss, (pg, bl, ent) = tr.impala_loss(lt.reshape(T1, Bm, A), vt.reshape(T1, Bm), mu, actions, rewards, dones, first)
    loss.backward()
    np.testing.assert_allclose(stats, [loss.item(), pg.item(), bl.item(), ent.item()], rtol=2e-5, atol=2e-6)
    g = flat.grad.numpy()
    np.testing.assert_allclose(grads, g, rtol=0, atol=1e-5 * np.abs(g).max())


def test_vtrace_on_policy_equals_gae_lambda1(oracle):
    # pi == mu => rho = 1 => errors are lambda=1 GAE advantages (SURVEY §8c analytic pin)
    rng = np.random.default_rng(1)
    T, B = 12, 4
    v = rng.normal(size=(T + 1, B)).astype(np.float32)
    r = rng.normal(size=(T, B)).astype(np.float32)
    d = (rng.random((T + 1, B)) < 0.2)
    disc = ((1 - d[:T]) * 0.99).astype(np.float32)
    errors, pg, q = oracle.vtrace(v[:-1], v[1:], r, disc, np.ones((T, B), np.float32))
    adv = np.zeros(B)
    ref = np.zeros((T, B))
    for t in reversed(range(T)):
        adv = r[t] + disc[t] * v[t + 1] - v[t] + disc[t] * adv
        ref[t] = adv
    np.testing.assert_allclose(errors, ref, rtol=0, atol=5e-6)
    np.testing.assert_allclose(pg, ref, rtol=0, atol=5e-6)   # rho=1: pg_adv == errors


def test_gae_geometric_series_and_advnorm(oracle):
    T, B = 128, 8
    r = np.ones((T, B), np.float32)
    z = np.zeros((T, B), np.float32)
    adv, tgt = oracle.gae(r, z, z.astype(np.uint8), np.zeros(B, np.float32), np.zeros(B, np.uint8))
    gl = 0.99 * 0.95
    ref = np.array([(1 - gl ** (T - t)) / (1 - gl) for t in range(T)])
    np.testing.assert_allclose(adv[:, 0], ref, rtol=2e-6)
    assert (tgt == adv).all()
    # a done at t+1 cuts the bootstrap
    d = z.astype(np.uint8).copy(); d[64] = 1
    adv2, _ = oracle.gae(r, z, d, np.zeros(B, np.float32), np.zeros(B, np.uint8))
    assert abs(adv2[63, 0] - 1.0) < 1e-7
    x = np.random.default_rng(0).normal(2, 3, size=(T, B)).astype(np.float32)
    n = oracle.advnorm(x, 4)
    g = x.reshape(T, 4, B // 4).astype(np.float64)
    ref = ((g - g.mean((0, 2), keepdims=True)) / (g.std((0, 2), keepdims=True) + 1e-8)).reshape(T, B)
    np.testing.assert_allclose(n, ref, rtol=0, atol=2e-6)


def test_adam_and_rmsprop_vs_torch(oracle):
    rng = np.random.default_rng(2)
    n = 1000
    p0 = rng.normal(size=n).astype(np.float32)
    for scale in (1e-4, 10.0):            # below and above the clip threshold
        p = p0.copy(); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        pt = torch.tensor(p0.astype(np.float64), requires_grad=True)
        opt = torch.optim.Adam([pt], lr=2.5e-4, eps=1e-5)
        for step in range(1, 4):
            g = (scale * rng.normal(size=n)).astype(np.float32)
            oracle.adam_step(p, g, m, v, 0.5, 2.5e-4, count=step)
            gn = np.sqrt((g.astype(np.float64) ** 2).sum())
            pt.grad = torch.tensor(g.astype(np.float64) * (1.0 if gn < 0.5 else 0.5 / gn))
            opt.step()
        np.testing.assert_allclose(p, pt.detach().numpy(), rtol=0, atol=2e-6)
    p = p0.copy(); nu = np.zeros(n, np.float32)
    pt = torch.tensor(p0.astype(np.float64), requires_grad=True)
    opt = torch.optim.RMSprop([pt], lr=6e-4, alpha=0.99, eps=0.01)
    for step in range(3):
        g = rng.normal(size=n).astype(np.float32)
        oracle.rmsprop_step(p, g, nu, 40.0, 6e-4)
        pt.grad = torch.tensor(g.astype(np.float64)); opt.step()
    np.testing.assert_allclose(p, pt.detach().numpy(), rtol=0, atol=2e-6)


def test_sampling_distribution_and_logprob(oracle):
    rng = np.random.default_rng(5)
    B = 4000
    logits = np.tile(rng.normal(size=(1, A)).astype(np.float32), (B, 1))
    actions, lp, key2 = oracle.sample_actions(logits, oracle.prng_key(123))
    p = np.exp(logits[0] - logits[0].max()); p /= p.sum()
    freq = np.bincount(actions, minlength=A) / B
    assert np.abs(freq - p).max() < 0.03
    np.testing.assert_allclose(lp, np.log(p)[actions], rtol=0, atol=1e-6)
    assert list(key2) == list(oracle.split(oracle.prng_key(123), 2)[0])
