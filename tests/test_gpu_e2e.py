"""End-to-end GPU parity: a whole device-env rollout and a whole learner update through the C ABI,
replayed step by step with the host env twin + the CPU oracle.  Frames, actions, rewards, dones are
compared bit-for-bit; losses and post-update parameters within 1e-5."""
import os
import sys
import numpy as np
import pytest

import cleanba_amd.lib as L
import cleanba_amd.model as M
import cleanba_amd.prng as prng

pytestmark = pytest.mark.gpu
A = 18


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def _init(seed):
    key = prng.prng_key(seed)
    key, nk, ak, ck = prng.split(key, 4)
    return key, M.init_nature_params(A, nk, ak, ck)


def _sched(n, lr, count0=0):
    bc = [M.adam_bias_corrections(count0 + i + 1) for i in range(n)]
    return np.full(n, lr, np.float32), np.array([b[0] for b in bc], np.float32), np.array([b[1] for b in bc], np.float32)


def test_ppo_rollout_and_update_match_oracle(oracle):
    E, T, S = 8, 16, 2
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, S, T
    ctx = L.Context(cfg)
    B = E * S
    key, params = _init(1)
    ctx.set_params(params)
    for s in range(S):
        ctx.actor_set_key(s, key)
        ctx.actor_env_reset_device(s, 5 + s)
        ctx.actor_begin_rollout(s, False)
        ctx.actor_rollout_device(s, T)
        ctx.actor_commit(s)
    ctx.learner_wait()
    obs = ctx.read("obs", np.uint8).reshape(T + 1, B, 4, 84, 84)
    actions = ctx.read("actions", np.int32).reshape(T + 1, B)[:T]
    logprobs = ctx.read("logprobs", np.float32).reshape(T + 1, B)[:T]
    values = ctx.read("values", np.float32).reshape(T + 1, B)[:T]
    rewards = ctx.read("rewards", np.float32).reshape(T + 1, B)[:T]
    dones = ctx.read("dones", np.uint8).reshape(T + 1, B)

    # ---- replay the rollout: host env twin + oracle policy (same key chain, ppo:256)
    for s in range(S):
        cols = slice(s * E, (s + 1) * E)
        st, o = L.synth_env_reset_host(5 + s, E)
        k = key.copy()
        assert (dones[0, cols] == 0).all()
        for t in range(T):
            assert (o == obs[t, cols]).all(), f"frames differ at t={t}"
            logits, value = oracle.nature_forward(params, A, o, ksplit=cfg.actor_dense_ksplit)
            a, lp, k = oracle.sample_actions(logits, k)
            assert (a == actions[t, cols]).all(), f"sampled actions differ at t={t}"
            assert (bits(lp) == bits(logprobs[t, cols])).all() and (bits(value) == bits(values[t, cols])).all()
            r, d, _, _ = L.synth_env_step_host(5 + s, st, o, a)
            assert (r == rewards[t, cols]).all() and (d == dones[t + 1, cols]).all()
        assert (o == obs[T, cols]).all()
        assert (ctx.actor_get_key(s) == k).all()

    # ---- learner update vs oracle (ppo:579-654)
    lkey = prng.prng_key(77)
    lrs, bc1, bc2 = _sched(16, 2.5e-4)
    key_after, stats = ctx.learner_update(lkey, lrs, bc1, bc2)
    p_gpu = ctx.get_params()
    adv_gpu = ctx.read("adv", np.float32).reshape(T + 1, B)[:T]

    _, nv = oracle.nature_forward(params, A, obs[T], ksplit=cfg.actor_dense_ksplit)
    adv, tgt = oracle.gae(rewards, values, dones[:T], nv, dones[T])
    adv = oracle.advnorm(adv, 4)
    np.testing.assert_allclose(adv_gpu, adv, rtol=0, atol=2e-5)
    N, MB = T * B, T * B // 4
    flat_obs = obs[:T].reshape(N, 4, 84, 84)
    fa, fl, fadv, ftgt = actions.reshape(N), logprobs.reshape(N), adv.reshape(N), tgt.reshape(N)
    p = params.copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    k = lkey.copy()
    ref_stats = []
    step = 0
    for e in range(4):
        ks = oracle.split(k, 2); k, sub = ks[0], ks[1]
        perm = oracle.permutation(sub, N)
        for mb in range(4):
            idx = perm[mb * MB:(mb + 1) * MB]
            st5, g, _, _ = oracle.ppo_loss_grad(p, A, flat_obs, idx, fa[idx], fl[idx], fadv[idx], ftgt[idx])
            oracle.adam_step(p, g, m, v, 0.5, float(lrs[step]), bc1=float(bc1[step]), bc2=float(bc2[step]))
            ref_stats.append(st5); step += 1
    assert (key_after == k).all()
    ref_stats = np.array(ref_stats)
    np.testing.assert_allclose(stats, ref_stats, rtol=2e-4, atol=2e-5)
    assert np.abs(p_gpu - p).max() <= 1e-5 * np.abs(p).max(), np.abs(p_gpu - p).max()
    assert abs(np.abs(p_gpu).sum() - np.abs(p).sum()) <= 1e-5 * np.abs(p).sum()
    ctx.close()


@pytest.mark.parametrize("A2,E", [(4, 5), (6, 3), (17, 7), (27, 2)])
def test_other_action_counts_match_oracle(oracle, A2, E):
    """num_actions other than 18 (envpool games with smaller action sets, ppo:135 full_action_space=False) and odd env counts: the per-frame
    actor tail (heads on one live MFMA row, A + 1 <= 32 columns over two waves), the fused env step and one learner update against the oracle."""
    T = 6
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions = E, 1, T, A2
    cfg.num_minibatches, cfg.update_epochs = 1, 1
    ctx = L.Context(cfg)
    key = prng.prng_key(9)
    key, nk, ak, ck = prng.split(key, 4)
    params = M.init_nature_params(A2, nk, ak, ck)
    ctx.set_params(params)
    ctx.actor_set_key(0, key)
    ctx.actor_env_reset_device(0, 3)
    ctx.actor_begin_rollout(0, False)
    ctx.actor_rollout_device(0, T)
    ctx.actor_commit(0)
    ctx.learner_wait()
    obs = ctx.read("obs", np.uint8).reshape(T + 1, E, 4, 84, 84)
    actions = ctx.read("actions", np.int32).reshape(T + 1, E)[:T]
    logprobs = ctx.read("logprobs", np.float32).reshape(T + 1, E)[:T]
    values = ctx.read("values", np.float32).reshape(T + 1, E)[:T]
    rewards = ctx.read("rewards", np.float32).reshape(T + 1, E)[:T]
    dones = ctx.read("dones", np.uint8).reshape(T + 1, E)
    st, o = L.synth_env_reset_host(3, E)
    k = key.copy()
    for t in range(T):
        assert (o == obs[t]).all(), f"frames differ at t={t}"
        logits, value = oracle.nature_forward(params, A2, o, ksplit=cfg.actor_dense_ksplit)
        a, lp, k = oracle.sample_actions(logits, k)
        assert (a == actions[t]).all() and a.max() < A2, f"sampled actions differ at t={t}"
        assert (bits(lp) == bits(logprobs[t])).all() and (bits(value) == bits(values[t])).all()
        r, d, _, _ = L.synth_env_step_host(3, st, o, a)
        assert (r == rewards[t]).all() and (d == dones[t + 1]).all()
    assert (o == obs[T]).all()
    lkey = prng.prng_key(5)
    lrs, bc1, bc2 = _sched(1, 2.5e-4)
    key_after, stats = ctx.learner_update(lkey, lrs, bc1, bc2)
    p_gpu = ctx.get_params()
    _, nv = oracle.nature_forward(params, A2, obs[T], ksplit=cfg.actor_dense_ksplit)
    adv, tgt = oracle.gae(rewards, values, dones[:T], nv, dones[T])
    adv = oracle.advnorm(adv, 1)
    N = T * E
    ks = oracle.split(lkey.copy(), 2)
    perm = oracle.permutation(ks[1], N)
    p = params.copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    st5, g, _, _ = oracle.ppo_loss_grad(p, A2, obs[:T].reshape(N, 4, 84, 84), perm, actions.reshape(N)[perm], logprobs.reshape(N)[perm],
                                        adv.reshape(N)[perm], tgt.reshape(N)[perm])
    oracle.adam_step(p, g, m, v, 0.5, float(lrs[0]), bc1=float(bc1[0]), bc2=float(bc2[0]))
    assert (key_after == ks[0]).all()
    np.testing.assert_allclose(stats[0], st5, rtol=2e-4, atol=2e-5)
    assert np.abs(p_gpu - p).max() <= 1e-5 * np.abs(p).max()
    ctx.close()


def test_impala_rollouts_and_update_match_oracle(oracle):
    E, T = 8, 6
    cfg = L.default_config(L.ALGO_IMPALA)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, 1, T
    ctx = L.Context(cfg)
    key, params = _init(2)
    ctx.set_params(params)
    ctx.actor_set_key(0, key)
    ctx.actor_env_reset_device(0, 9)
    st, o = L.synth_env_reset_host(9, E)
    k = key.copy()
    rew = np.zeros(E, np.float32); dn = np.zeros(E, np.uint8); fs = np.ones(E, np.uint8)
    carried = None
    p = params.copy(); nu = np.zeros_like(p)
    for u in (1, 2):
        ctx.actor_begin_rollout(0, True)
        ctx.actor_rollout_device(0, T + 1 if u == 1 else T)
        ctx.actor_commit(0)
        ctx.learner_wait()
        r = u - 1
        obs = ctx.read("obs", np.uint8, ring=r).reshape(T + 1, E, 4, 84, 84)
        actions = ctx.read("actions", np.int32, ring=r).reshape(T + 1, E)
        mu = ctx.read("logits", np.float32, ring=r).reshape(T + 1, E, A)
        rewards = ctx.read("rewards", np.float32, ring=r).reshape(T + 1, E)
        dones = ctx.read("dones", np.uint8, ring=r).reshape(T + 1, E)
        firsts = ctx.read("firststeps", np.uint8, ring=r).reshape(T + 1, E)
        t0 = 0
        if carried is not None:   # storage = storage[-1:]  (impala:416)
            for a_, b_ in zip(carried, (obs[0], actions[0], mu[0], rewards[0], dones[0], firsts[0])):
                assert (np.asarray(a_) == b_).all()
            t0 = 1
        # actor params: version 0 for both rollouts under --concurrency (ppo:287-304)
        for t in range(t0, T + 1):
            assert (o == obs[t]).all() and (rew == rewards[t]).all() and (dn == dones[t]).all() and (fs == firsts[t]).all()
            logits, _ = oracle.nature_forward(params, A, o, ksplit=cfg.actor_dense_ksplit)
            a, _, k = oracle.sample_actions(logits, k)
            assert (a == actions[t]).all() and (bits(logits) == bits(mu[t])).all()
            if t < T:
                rew, d, _, el = L.synth_env_step_host(9, st, o, a)
                dn = d; fs = (el == 0).astype(np.uint8)
        carried = (obs[T].copy(), actions[T].copy(), mu[T].copy(), rewards[T].copy(), dones[T].copy(), firsts[T].copy())
        # the carried action is sent to the env at the start of the next rollout
        rew, d, _, el = L.synth_env_step_host(9, st, o, actions[T])
        dn = d; fs = (el == 0).astype(np.uint8)

        lrs = np.full(4, 6e-4, np.float32)
        _, stats = ctx.learner_update(prng.prng_key(0), lrs, np.ones(4, np.float32), np.ones(4, np.float32))
        Bm = E // 4
        ref = []
        for mb in range(4):
            c = slice(mb * Bm, (mb + 1) * Bm)
            st4, g = oracle.impala_loss_grad(p, A, obs[:, c].reshape(-1, 4, 84, 84), None, T + 1, Bm, mu[:, c], actions[:, c], rewards[:, c],
                                             dones[:, c], firsts[:, c])
            oracle.rmsprop_step(p, g, nu, 40.0, 6e-4)
            ref.append(st4)
        np.testing.assert_allclose(stats, np.array(ref), rtol=2e-4, atol=2e-4)
        p_gpu = ctx.get_params()
        assert np.abs(p_gpu - p).max() <= 2e-5 * np.abs(p).max(), np.abs(p_gpu - p).max()
    ctx.close()


def test_atari57_mix_device_env_matches_host_twin():
    # configs[4]: the device env renders the 57-preset mix straight into the ring; the host twin must produce the same bytes
    E, T = 64, 12
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, 1, T
    ctx = L.Context(cfg)
    key, params = _init(3)
    ctx.set_params(params)
    ctx.actor_set_key(0, key)
    ctx.actor_env_reset_device(0, 11, atari57_mix=True)
    ctx.actor_begin_rollout(0, False)
    ctx.actor_rollout_device(0, T)
    ctx.actor_commit(0)
    ctx.learner_wait()
    obs = ctx.read("obs", np.uint8).reshape(T + 1, E, 4, 84, 84)
    actions = ctx.read("actions", np.int32).reshape(T + 1, E)[:T]
    rewards = ctx.read("rewards", np.float32).reshape(T + 1, E)[:T]
    dones = ctx.read("dones", np.uint8).reshape(T + 1, E)
    st, o = L.synth_env_reset_host(11, E, atari57_mix=True)
    for t in range(T):
        assert (o == obs[t]).all(), f"frames differ at t={t}"
        r, d, _, _ = L.synth_env_step_host(11, st, o, actions[t])
        assert (r == rewards[t]).all() and (d == dones[t + 1]).all()
    assert (o == obs[T]).all()
    assert len({obs[0, e, 3].tobytes() for e in range(57)}) > 50
    ctx.close()


@pytest.mark.parametrize("ksplit", [14, 49])
def test_device_env_long_rollout_matches_host_twin_through_rewards_and_resets(ksplit):
    """The device env steps in pieces around the action (env_model.h: three precomputed transitions, the new plane = the previous one with the ball
    moved, paddle rows and changed bricks after the action).  300 steps of 114 envs (two of each preset) see dozens of rewards, brick changes and
    episode resets; every byte of every stack must equal the host twin's, which paints whole planes.  ksplit = 14: the env step fused into the
    per-frame actor tail; 49: the stand-alone env_step_kernel."""
    E, T = 114, 300
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_minibatches, cfg.ring_depth, cfg.actor_dense_ksplit = E, 1, T, 6, 2, ksplit
    ctx = L.Context(cfg)
    try:
        key, params = _init(5)
        ctx.set_params(params)
        ctx.actor_set_key(0, key)
        ctx.actor_env_reset_device(0, 23, atari57_mix=True)
        ctx.actor_begin_rollout(0, False)
        ctx.actor_rollout_device(0, T)
        ctx.actor_commit(0)
        ctx.learner_wait()
        obs = ctx.read("obs", np.uint8).reshape(T + 1, E, 4, 84, 84)
        actions = ctx.read("actions", np.int32).reshape(T + 1, E)[:T]
        rewards = ctx.read("rewards", np.float32).reshape(T + 1, E)[:T]
        dones = ctx.read("dones", np.uint8).reshape(T + 1, E)
        st, o = L.synth_env_reset_host(23, E, atari57_mix=True)
        n_done = n_reward = 0
        for t in range(T):
            assert (o == obs[t]).all(), f"frames differ at t={t}: envs {np.nonzero((o != obs[t]).reshape(E, -1).any(axis=1))[0][:8]}"
            r, d, _, _ = L.synth_env_step_host(23, st, o, actions[t])
            assert (r == rewards[t]).all() and (d == dones[t + 1]).all(), t
            n_done += int(d.sum())
            n_reward += int((r > 0).sum())
        assert (o == obs[T]).all()
        print(f"device env vs host twin: {n_reward} rewards, {n_done} episode ends in {E * T} env-steps")
        assert n_done >= 5 and n_reward >= 100
    finally:
        ctx.close()


@pytest.mark.parametrize("algo", ["ppo", "impala"])
def test_gradient_accumulation_update_matches_oracle(oracle, algo):
    # optax.MultiSteps(every_k_schedule=2): the library's fused update vs the oracle-backed engine replaying the same ring
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    E, T, S = 8, 8, 2
    cfg = L.default_config(L.ALGO_PPO if algo == "ppo" else L.ALGO_IMPALA)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, S, T
    cfg.num_minibatches, cfg.grad_accum_steps, cfg.update_epochs = 2, 2, 2
    ctx = L.Context(cfg)
    key, params = _init(4)
    ctx.set_params(params)
    for s in range(S):
        ctx.actor_set_key(s, key)
        ctx.actor_env_reset_device(s, 21 + s)
        ctx.actor_begin_rollout(s, False)
        ctx.actor_rollout_device(s, T + (0 if algo == "ppo" else 1))
        ctx.actor_commit(s)
    ctx.learner_wait()
    ora = OracleEngine(cfg)
    ora.set_params(params)
    B = E * S
    R = ora.ring[0]
    R["obs"][:] = ctx.read("obs", np.uint8).reshape(T + 1, B, 4, 84, 84)
    for name, dt in (("actions", np.int32), ("logprobs", np.float32), ("values", np.float32), ("rewards", np.float32), ("dones", np.uint8),
                     ("firststeps", np.uint8)):
        R[name][:] = ctx.read(name, dt).reshape(T + 1, B)
    R["logits"][:] = ctx.read("logits", np.float32).reshape(T + 1, B, A)
    ora.committed = [1] * S
    n_opt = 2 * (2 if algo == "ppo" else 1)
    lrs, bc1, bc2 = _sched(n_opt, 1e-3)
    k1, stats = ctx.learner_update(key, lrs, bc1, bc2)
    k2, stats_o = ora.learner_update(key, lrs, bc1, bc2)
    assert stats.shape == stats_o.shape == (n_opt * 2, 5 if algo == "ppo" else 4)
    np.testing.assert_allclose(stats, stats_o, rtol=2e-4, atol=2e-5)
    p_gpu, p_ora = ctx.get_params(), ora.get_params()
    assert np.abs(p_gpu - p_ora).max() <= 1e-5 * max(1.0, np.abs(p_ora).max())
    assert (k1 == k2).all()
    ctx.close()


def test_save_model_then_evaluate_matches_oracle_replay(tmp_path, oracle):
    # ppo:753-785: --save-model writes the .cleanrl_model and runs the evaluation loop (eval.py:13-82); replay it with the oracle
    import os
    from cleanba_amd.args import parse_args
    from cleanba_amd.checkpoint import load_cleanrl_model
    from cleanba_amd.envs import make_env
    from cleanba_amd.trainer import train
    os.chdir(str(tmp_path))
    args = parse_args(["--local-num-envs", "8", "--num-actor-threads", "1", "--num-steps", "8", "--env-backend", "device", "--network", "nature",
                       "--total-timesteps", "128", "--save-model", "--eval-episodes", "2", "--eval-max-episode-steps", "48"], "ppo")
    res = train(args, "ppo")
    assert os.path.exists(res["model_path"]) and len(res["eval_returns"]) == 2
    _, params = load_cleanrl_model(res["model_path"], 18, "nature")
    assert np.array_equal(params, res["params"])
    envs = make_env(args.env_id, 1, 1, backend="host")()
    key = prng.split(prng.prng_key(1), 4)[0]
    want = []
    for ep in range(2):
        obs, ret = envs.reset(), 0.0
        for _ in range(48):
            logits, _ = oracle.nature_forward(params, 18, obs, ksplit=14)
            a, _, key = oracle.sample_actions(logits, key)
            obs, _, _, info = envs.step(a)
            ret += float(info["reward"][0])
            if int(info["terminated"].sum()) + int(info["TimeLimit.truncated"].sum()) >= 1:
                break
        want.append(ret)
    assert res["eval_returns"] == want


@pytest.mark.parametrize("algo", ["ppo", "impala"])
def test_host_env_loop_equals_device_env_loop(tmp_path, algo):
    # the envpool-style loop (cbm_actor_step_host / record_host / commit(next_obs), ppo:308-375, impala:351-416) against the device
    # env: the two envs are byte-identical twins, so the whole run must agree bit for bit
    import os
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train
    os.chdir(str(tmp_path))
    out = {}
    for backend in ("device", "host"):
        args = parse_args(["--local-num-envs", "8", "--num-actor-threads", "2", "--num-steps", "8", "--env-backend", backend, "--network", "nature",
                           "--total-timesteps", str(3 * 8 * 2 * 8), "--log-frequency", "1000", "--update-epochs", "2"], algo)
        out[backend] = train(args, algo)
    assert out["host"]["updates"] == out["device"]["updates"] == 3
    assert np.array_equal(out["host"]["params"], out["device"]["params"])


def test_page_locked_host_buffer_gives_the_same_actions():
    """cbm_host_register: the env's observation buffer page-locked (DMA from the caller's pages) vs pageable (runtime staging) — same uploads,
    same actions, same stored rows."""
    import cleanba_amd.lib as L
    import cleanba_amd.model as M
    import cleanba_amd.prng as prng
    E, T = 12, 6
    outs = []
    for pinned in (False, True):
        cfg = L.default_config(L.ALGO_PPO)
        cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, 1, T
        ctx = L.Context(cfg)
        key = prng.prng_key(3)
        key, nk, ak, ck = prng.split(key, 4)
        ctx.set_params(M.init_nature_params(18, nk, ak, ck))
        ctx.actor_set_key(0, key)
        st, obs = L.synth_env_reset_host(7, E)
        if pinned:
            ctx.host_register(obs)
        done = np.zeros(E, np.uint8)
        ctx.actor_begin_rollout(0, False)
        ring = ctx.actor_ring_index(0)
        acts = []
        for t in range(T):
            a = ctx.actor_step_host(0, obs, done)
            acts.append(a.copy())
            r, d, _, _ = L.synth_env_step_host(7, st, obs, a)      # in place: the registered buffer is reused
            ctx.actor_record_host(0, r)
            done = d
        ctx.actor_commit(0, obs, done)
        ctx.sync()
        stored = ctx.read("obs", np.uint8, ring=ring)[:T * E * 28224].copy()
        if pinned:
            ctx.host_unregister(obs)
        ctx.close()
        outs.append((np.stack(acts), stored))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("algo,E,thr,T,nmb,epochs,accum,conc", [
    ("ppo", 2, 1, 4, 1, 1, 1, False), ("ppo", 3, 2, 5, 2, 3, 1, True), ("ppo", 4, 3, 4, 4, 2, 2, True),
    ("impala", 2, 2, 4, 2, 1, 1, False), ("impala", 6, 1, 5, 3, 1, 2, True), ("impala", 4, 3, 3, 4, 1, 1, True)])
def test_whole_training_runs_match_the_oracle_engine(tmp_path, algo, E, thr, T, nmb, epochs, accum, conc):
    # the same host program (cleanba_amd.trainer.train) once on the HIP library and once on the oracle-backed CPU engine, odd shapes:
    # actor threads, ring hand-off, policy-version skew, minibatching, accumulation, schedules.  Actions are bit-exact, so the runs see
    # the same data; parameters may differ by the backward tolerance.
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train
    os.chdir(str(tmp_path))
    updates = 4
    argv = ["--local-num-envs", str(E), "--num-actor-threads", str(thr), "--num-steps", str(T), "--num-minibatches", str(nmb),
            "--update-epochs", str(epochs), "--gradient-accumulation-steps", str(accum), "--network", "nature", "--env-backend", "host",
            "--total-timesteps", str(updates * E * thr * T), "--log-frequency", "1000"] + (["--concurrency"] if conc else [])
    gpu = train(parse_args(argv, algo), algo)
    cpu = train(parse_args(argv, algo), algo, engine_factory=OracleEngine)
    assert gpu["updates"] == cpu["updates"] == updates
    d = np.abs(gpu["params"] - cpu["params"]).max()
    assert d <= 2e-5 * max(1.0, np.abs(cpu["params"]).max()), d
    np.testing.assert_allclose(gpu["stats"], cpu["stats"], rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("E,Ba,T,nmb,epochs,conc", [(4, 2, 4, 2, 2, False), (6, 2, 5, 3, 1, True), (12, 4, 6, 4, 2, True)])
def test_async_batch_size_runs_match_the_oracle_engine(tmp_path, E, Ba, T, nmb, epochs, conc):
    # legacy `--async-batch-size` (legacy_scripts/..._naturecnn.py): recv() returns Ba of the E envs in completion order, the
    # rollout carries env ids, returns are env-id-indexed, advantages normalised per minibatch.  Same host program on the HIP library and
    # on the oracle-backed engine; sampled actions are bit-exact, so both see the same env-id pattern.
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train
    os.chdir(str(tmp_path))
    updates = 3
    argv = ["--local-num-envs", str(E), "--async-batch-size", str(Ba), "--num-steps", str(T), "--num-minibatches", str(nmb), "--update-epochs",
            str(epochs), "--network", "nature", "--env-backend", "host", "--total-timesteps", str(updates * E * T), "--log-frequency", "1000"] + (
                ["--concurrency"] if conc else [])
    gpu = train(parse_args(argv, "ppo"), "ppo")
    cpu = train(parse_args(argv, "ppo"), "ppo", engine_factory=OracleEngine)
    assert gpu["updates"] == cpu["updates"] == updates
    d = np.abs(gpu["params"] - cpu["params"]).max()
    assert d <= 2e-5 * max(1.0, np.abs(cpu["params"]).max()), d
    np.testing.assert_allclose(gpu["stats"], cpu["stats"], rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("split", [2, 3])
def test_backward_split_training_run_matches_the_oracle_engine(tmp_path, split):
    # `--backward-split` (build-only extension): whole PPO runs with the backward GEMMs on split-bf16 MFMA against the fp32 oracle engine.
    # Sampled actions stay bit-exact because the forward pass is untouched.  Three-term splits hold the fp32 runs' bound (measured 1e-7, the same
    # as the fp32 MFMA path); two-term splits carry a ~1e-6 relative gradient error that Adam's normalisation of small elements turns into
    # 3e-5 after 16 steps (0.8 % of the distance the parameters moved) — bounded here at 1e-4.
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train
    os.chdir(str(tmp_path))
    updates = 4
    argv = ["--local-num-envs", "4", "--num-actor-threads", "2", "--num-steps", "5", "--num-minibatches", "2", "--update-epochs", "2",
            "--network", "nature", "--env-backend", "host", "--total-timesteps", str(updates * 4 * 2 * 5), "--log-frequency", "1000", "--concurrency"]
    gpu = train(parse_args(argv + ["--backward-split", str(split)], "ppo"), "ppo")
    cpu = train(parse_args(argv, "ppo"), "ppo", engine_factory=OracleEngine)
    assert gpu["updates"] == cpu["updates"] == updates
    d = np.abs(gpu["params"] - cpu["params"]).max()
    assert d <= (2e-5 if split == 3 else 1e-4) * max(1.0, np.abs(cpu["params"]).max()), d
    np.testing.assert_allclose(gpu["stats"], cpu["stats"], rtol=5e-4, atol=5e-5)


@pytest.mark.gpu
def test_dataflow_actor_step_experiment_is_bit_exact():
    """CBM_ACTOR_FUSED=1 (gemm_layers.hip actor_fused_kernel: conv1 .. dense of an actor step as ONE launch, blocks waiting on per-frame arrival
    counters instead of kernel boundaries) is a measured experiment that does not ship (profiles/r06_actor_dataflow.txt: 2.1x slower) — but while it is
    in the tree it has to produce the oracle's rollouts: the switch is read once per process, so the rollout parity tests run again in a child."""
    import subprocess
    env = dict(os.environ, CBM_ACTOR_FUSED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "test_ppo_rollout_and_update_match_oracle or test_impala_rollouts_and_update_match_oracle or host_env_loop_equals_device_env_loop"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
