"""Synthetic env: host twin semantics (envpool-like auto-reset, frame stacking, sparsity)."""
import numpy as np
import pytest

import cleanba_amd.lib as L


def test_host_env_semantics():
    n = 16
    st, obs = L.synth_env_reset_host(3, n)
    assert obs.shape == (n, 4, 84, 84) and (obs[:, 0] == obs[:, 3]).all()
    frac_zero = (obs == 0).mean()
    assert 0.75 < frac_zero < 0.95
    rng = np.random.default_rng(0)
    tot_r, tot_d, steps = 0.0, 0, 3000
    prev = obs.copy()
    was_done = np.zeros(n, bool)
    for t in range(steps):
        a = rng.integers(0, 18, n).astype(np.int32)
        r, d, term, el = L.synth_env_step_host(3, st, obs, a)
        # frame stack shifts left unless the env was reset this step
        keep = ~was_done
        assert (obs[keep, :3] == prev[keep, 1:]).all()
        # step after done: reset -> elapsed 0, reward 0, not done, stack filled with the first frame
        assert (el[was_done] == 0).all() and (r[was_done] == 0).all() and (d[was_done] == 0).all()
        assert (obs[was_done, 0] == obs[was_done, 3]).all()
        assert set(np.unique(r)).issubset({0.0, 1.0})
        tot_r += r.sum(); tot_d += d.sum()
        was_done = d.astype(bool)
        prev = obs.copy()
    assert 0.01 < tot_r / (steps * n) < 0.03 and 1 / 1600 < tot_d / (steps * n) < 1 / 400


def test_env_is_deterministic_and_action_dependent():
    def run(seed, acts):
        st, obs = L.synth_env_reset_host(seed, 4)
        out = []
        for a in acts:
            r, d, _, _ = L.synth_env_step_host(seed, st, obs, np.full(4, a, np.int32))
            out.append((obs.copy(), r.copy()))
        return out
    a = run(5, [1] * 20); b = run(5, [1] * 20); c = run(5, [2] * 20)
    assert all((x[0] == y[0]).all() for x, y in zip(a, b))
    assert any((x[0] != y[0]).any() for x, y in zip(a, c))


def test_atari57_mix_presets():
    # BASELINE configs[4] "synthetic Atari-57 frame mix": env e plays preset e % 57; preset 0 is Breakout (the plain env)
    n = 114
    st0, obs0 = L.synth_env_reset_host(5, n)
    st, obs = L.synth_env_reset_host(5, n, atari57_mix=True)
    assert [s.game for s in st] == [e % 57 for e in range(n)] and all(s.game == 0 for s in st0)
    assert (obs[0] == obs0[0]).all() and (obs[57] == obs0[57]).all()          # game 0 == Breakout preset, bit for bit
    first = obs[:57, 3].reshape(57, -1)
    assert len({f.tobytes() for f in first}) > 50                             # the presets really look different
    sparsity = (first == 0).mean(1)
    assert sparsity.min() > 0.7 and sparsity.max() - sparsity.min() > 0.05
    rng = np.random.default_rng(1)
    steps, rew, done = 1500, np.zeros(n), np.zeros(n)
    for t in range(steps):
        a = rng.integers(0, 18, n).astype(np.int32)
        r, d, _, _ = L.synth_env_step_host(5, st, obs, a)
        r0, d0, _, _ = L.synth_env_step_host(5, st0, obs0, a)
        rew += r; done += d
        assert (obs[0] == obs0[0]).all() and r[0] == r0[0] and d[0] == d0[0]
    # same game, different env id -> different event stream; reward / episode-length rates spread over the presets
    rate = (rew[:57] + rew[57:]) / (2 * steps)
    assert 0.002 < rate.min() and rate.max() < 0.07 and rate.max() > 2 * rate.min()
    assert done.sum() > 0


def test_async_batches_are_the_same_trajectories_in_a_different_order():
    """envpool async mode (batch_size < num_envs): recv() hands back batch_size envs in completion order.  Whatever the
    order, every env must live exactly the trajectory it has in the synchronous env when it is fed the same actions — observation,
    and the reward / done / elapsed_step that arrive WITH that observation."""
    import zlib
    from cleanba_amd.envs import SyntheticAtariEnv
    E, Ba, steps = 12, 4, 40
    policy = lambda e, k: (e * 5 + k * 3) % 6
    sync = SyntheticAtariEnv(num_envs=E, seed=9)
    obs = sync.reset()
    want = [[(zlib.crc32(obs[e].tobytes()), 0.0, False, 0)] for e in range(E)]
    for k in range(steps):
        obs, r, d, info = sync.step(np.array([policy(e, k) for e in range(E)], np.int32))
        for e in range(E):
            want[e].append((zlib.crc32(obs[e].tobytes()), float(r[e]), bool(d[e]), int(info["elapsed_step"][e])))
    asy = SyntheticAtariEnv(num_envs=E, seed=9, batch_size=Ba)
    asy.async_reset()
    seen = [0] * E
    orders = set()
    buf = np.zeros(Ba, np.int32)
    while min(seen) < steps:
        o, r, d, info = asy.recv()
        ids = info["env_id"]
        assert len(set(ids.tolist())) == Ba
        orders.add(tuple(ids.tolist()))
        live = []
        for j, e in enumerate(ids):
            if seen[e] <= steps:
                assert (zlib.crc32(o[j].tobytes()), float(r[j]), bool(d[j]), int(info["elapsed_step"][j])) == want[e][seen[e]], (e, seen[e])
            buf[j] = policy(int(e), seen[e])
            seen[e] += 1
        asy.send(buf, ids)
        buf[:] = -7          # the caller may reuse its action buffer right after send()
    asy.close()
    assert len(orders) > 10   # the batches are not a fixed round-robin


def test_layered_host_painter_equals_the_per_pixel_function_and_out_of_place_step_equals_in_place():
    """The host twin paints a plane layer by layer and steps out of place (a fresh array per step like envpool's recv); the device kernels call
    the per-pixel function and shift in place.  Same bytes, for every game preset, through resets."""
    n, seed = 114, 11
    st_a, obs_a = L.synth_env_reset_host(seed, n, atari57_mix=True)
    st_b, obs_b = L.synth_env_reset_host(seed, n, atari57_mix=True)
    rng = np.random.default_rng(2)
    for t in range(400):
        a = rng.integers(0, 18, n).astype(np.int32)
        ra, da, ta, ea = L.synth_env_step_host(seed, st_a, obs_a, a, 200)          # in place (short episodes: many resets)
        obs_b, rb, db, tb, eb = L.synth_env_step_host_to(seed, st_b, obs_b, a, 200)   # out of place
        assert (obs_a == obs_b).all() and (ra == rb).all() and (da == db).all() and (ta == tb).all() and (ea == eb).all()
        if t % 25 == 0:
            for e in range(n):
                slow, fast = L.synth_env_render_host(st_a[e], 0), L.synth_env_render_host(st_a[e], 1)
                assert (slow == fast).all(), (t, e)
                assert (fast == obs_a[e, 3]).all()
                assert (L.synth_env_render_host(st_a[e], 2) == slow).all(), (t, e)   # env_word: what the device kernels paint with


def test_word_painter_equals_the_per_pixel_function_on_random_states():
    """env_word (the device kernels' painter: regions decided per 4-pixel word) against env_pixel on states the dynamics rarely reach: every ball
    position incl. the brick rows and the walls, paddles at both stops, sparse brick walls, all 57 presets."""
    rng = np.random.default_rng(5)
    st, _ = L.synth_env_reset_host(3, 57, atari57_mix=True)
    for rep in range(40):
        for e in range(57):
            s = st[e]
            s.ball_x, s.ball_y = int(rng.integers(1, 82)), int(rng.integers(12, 76))
            s.paddle_x = int(rng.choice([1, 36, 83 - 22, 83 - 8, int(rng.integers(1, 60))]))
            for w in range(3):
                s.bricks[w] = int(rng.integers(0, 1 << 28)) if rep % 3 else 0x0FFFFFFF
            assert (L.synth_env_render_host(s, 2) == L.synth_env_render_host(s, 0)).all(), (rep, e)
