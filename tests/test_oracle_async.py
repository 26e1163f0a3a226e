"""Async (`--async-batch-size`) returns of the legacy script, naturecnn:232-262 + 467-531.

The oracle restates the reference's row scan; it is pinned here by an INDEPENDENT formulation: pull out every env's own sample sequence
and run the textbook GAE recursion on it, with the last sample of the rollout having advantage 0 (no bootstrap observation in async mode).
"""
import numpy as np

import oracle


def make_async_rollout(R, B, NE, seed):
    """Rows of B distinct env ids in an irregular (latency-driven) order, every env stepped at least twice."""
    rng = np.random.default_rng(seed)
    env_ids = np.zeros((R, B), np.int32)
    ready = rng.random(NE)
    for r in range(R):
        pick = np.argsort(ready, kind="stable")[:B]
        env_ids[r] = np.sort(pick)
        ready[pick] = ready[pick].max() + rng.random(B) * 3.0
    rewards = (rng.random((R, B)) < 0.2).astype(np.float32)
    values = rng.standard_normal((R, B)).astype(np.float32)
    dones = (rng.random((R, B)) < 0.1).astype(np.uint8)
    return env_ids, rewards, values, dones


def per_env_gae(env_ids, rewards, values, dones, NE, gamma, lam):
    R, B = env_ids.shape
    adv = np.zeros(R * B, np.float64)
    flat = env_ids.reshape(-1)
    r, v, d = rewards.reshape(-1).astype(np.float64), values.reshape(-1).astype(np.float64), dones.reshape(-1).astype(np.float64)
    for e in range(NE):
        idx = np.nonzero(flat == e)[0]
        a = 0.0
        for k in range(len(idx) - 2, -1, -1):   # the env's last sample keeps advantage 0
            i, j = idx[k], idx[k + 1]
            nnt = 1.0 - d[j]                    # done / reward that arrived with the NEXT observation of this env
            delta = r[j] + gamma * v[j] * nnt - v[i]
            a = delta + gamma * lam * nnt * a
            adv[i] = a
    return adv.reshape(R, B), (adv + v).reshape(R, B)


def test_next_index_matches_definition():
    env_ids, *_ = make_async_rollout(40, 4, 12, 0)
    nxt = oracle.async_next_index(env_ids, 12)
    flat = env_ids.reshape(-1)
    for i, e in enumerate(flat):
        later = np.nonzero(flat[i + 1:] == e)[0]
        assert nxt[i] == (i + 1 + later[0] if len(later) else 0)


def test_async_gae_equals_per_env_gae():
    for (R, B, NE, seed) in [(60, 4, 12, 1), (384, 20, 60, 2), (33, 5, 5, 3), (16, 3, 9, 4)]:
        env_ids, rewards, values, dones = make_async_rollout(R, B, NE, seed)
        adv, tgt = oracle.gae_async(env_ids, rewards, values, dones, NE, 0.99, 0.95)
        ra, rt = per_env_gae(env_ids, rewards, values, dones, NE, 0.99, 0.95)
        np.testing.assert_allclose(adv, ra, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tgt, rt, rtol=1e-5, atol=1e-5)


def test_async_gae_reduces_to_sync_gae_for_round_robin():
    """With batch_size == num_envs every row is all envs: the async returns equal compute_gae (ppo:532-560) on rows 0..R-2 with
    next_value = values[R-1], shifted rewards/dones — and the last row is 0."""
    rng = np.random.default_rng(5)
    R, NE = 20, 6
    env_ids = np.tile(np.arange(NE, dtype=np.int32), (R, 1))
    rewards = rng.random((R, NE)).astype(np.float32)
    values = rng.standard_normal((R, NE)).astype(np.float32)
    dones = (rng.random((R, NE)) < 0.15).astype(np.uint8)
    adv, _ = oracle.gae_async(env_ids, rewards, values, dones, NE, 0.99, 0.95)
    ref, _ = oracle.gae(rewards[1:], values[:-1], dones[:-1], values[-1], dones[-1], 0.99, 0.95)
    np.testing.assert_array_equal(adv[:-1], ref)
    assert not adv[-1].any()


def test_mb_advnorm():
    x = np.random.default_rng(6).standard_normal(3840).astype(np.float32) * 3 + 1
    y = oracle.mb_advnorm(x)
    np.testing.assert_allclose(y, (x - x.mean()) / (x.std() + 1e-8), rtol=1e-5, atol=1e-6)
