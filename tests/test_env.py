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
