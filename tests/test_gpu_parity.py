"""GPU parity tests (run with -m gpu on an MI355X): every HIP entry point of the C ABI against the
CPU oracle on the same seeded inputs.  Bars (BASELINE.json north_star): sampled action indices
bit-exact, integer/index work bit-exact, logits/losses/gradients within 1e-5 (fp32).  Forward
logits are in fact bit-exact by construction (same fmaf chain order), which is asserted too."""
import numpy as np
import pytest

import cleanba_amd.lib as L
from helpers import make_frames, make_params

pytestmark = pytest.mark.gpu
A = 18


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ctx():
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 16, 1, 16   # MB = 64 frames of workspace
    c = L.Context(cfg)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ictx():
    cfg = L.default_config(L.ALGO_IMPALA)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 16, 1, 7
    c = L.Context(cfg)
    yield c
    c.close()


def test_forward_bit_exact(ctx, oracle):
    P = make_params(A, 11)
    obs = make_frames(40, 12)
    dP, dO = L.DevBuf(ctx, P), L.DevBuf(ctx, obs)
    for idx, ks in ((None, 1), (None, 14), ([7, 3, 39, 0, 3], 1), (list(range(39, 6, -1)), 14)):
        B = len(idx) if idx is not None else 40
        dI = L.DevBuf(ctx, np.asarray(idx, np.int32)) if idx is not None else None
        dL = L.DevBuf(ctx, nbytes=B * A * 4, dtype=np.float32, shape=(B, A))
        dV = L.DevBuf(ctx, nbytes=B * 4, dtype=np.float32, shape=(B,))
        L._chk(ctx.lib.cbm_forward(ctx.h, L._p(dP.ptr), L._p(dO.ptr), L._p(dI.ptr if dI else None), B, ks, L._p(dL.ptr), L._p(dV.ptr)))
        lo, vo = oracle.nature_forward(P, A, obs, idx=idx, ksplit=ks)
        lg, vg = dL.download(), dV.download()
        np.testing.assert_allclose(lg, lo, rtol=0, atol=1e-5)
        assert (bits(lg) == bits(lo)).all() and (bits(vg) == bits(vo)).all(), "forward must be bit-exact (same chain order)"


def test_sample_bit_exact(ctx, oracle):
    rng = np.random.default_rng(5)
    for B in (1, 16, 120, 121):
        logits = rng.normal(0, 1.5, size=(B, A)).astype(np.float32)
        key = oracle.prng_key(1234 + B)
        sub = oracle.split(key, 2)[1]
        dL = L.DevBuf(ctx, logits)
        dA = L.DevBuf(ctx, nbytes=B * 4, dtype=np.int32, shape=(B,))
        dP = L.DevBuf(ctx, nbytes=B * 4, dtype=np.float32, shape=(B,))
        L._chk(ctx.lib.cbm_sample(ctx.h, L._p(dL.ptr), B, L._p(np.ascontiguousarray(sub)), L._p(dA.ptr), L._p(dP.ptr)))
        a_o, lp_o, _ = oracle.sample_actions(logits, key)
        assert (dA.download() == a_o).all()
        assert (bits(dP.download()) == bits(lp_o)).all()


def test_gae_advnorm_permutation(ctx, oracle):
    rng = np.random.default_rng(6)
    T, B = 128, 120
    r = (rng.random((T, B)) < 0.05).astype(np.float32)
    v = rng.normal(size=(T, B)).astype(np.float32)
    d = (rng.random((T, B)) < 0.01).astype(np.uint8)
    nv = rng.normal(size=B).astype(np.float32)
    nd = (rng.random(B) < 0.05).astype(np.uint8)
    bufs = [L.DevBuf(ctx, x) for x in (r, v, d, nv, nd)]
    dA = L.DevBuf(ctx, nbytes=T * B * 4, dtype=np.float32, shape=(T, B))
    dT = L.DevBuf(ctx, nbytes=T * B * 4, dtype=np.float32, shape=(T, B))
    L._chk(ctx.lib.cbm_gae(ctx.h, *[L._p(b.ptr) for b in bufs], T, B, L._p(dA.ptr), L._p(dT.ptr)))
    adv_o, tgt_o = oracle.gae(r, v, d, nv, nd)
    adv = dA.download()
    assert (bits(adv) == bits(adv_o)).all() and (bits(dT.download()) == bits(tgt_o)).all()   # same serial order
    L._chk(ctx.lib.cbm_advnorm(ctx.h, L._p(dA.ptr), T, B, 4))
    np.testing.assert_allclose(dA.download(), oracle.advnorm(adv_o, 4), rtol=0, atol=1e-5)
    for n in (5, 1000, 15360, 30720):
        key = oracle.prng_key(n)
        dPm = L.DevBuf(ctx, nbytes=n * 4, dtype=np.int32, shape=(n,))
        L._chk(ctx.lib.cbm_permutation(ctx.h, L._p(np.ascontiguousarray(key)), n, L._p(dPm.ptr)))
        assert (dPm.download() == oracle.permutation(key, n)).all()
        dPm.free()


def test_vtrace_bit_exact(ctx, oracle):
    """rlax.vtrace_td_error_and_advantage (impala:559-567) as a pure function: same serial recursion as the oracle -> same bits; and at
    rho = 1 the errors are the lambda = 1 GAE advantages (the analytic pin of tests/test_oracle_network.py, here on the GPU)."""
    rng = np.random.default_rng(12)
    for T, B in ((20, 120), (128, 30), (1, 7), (1999, 3)):
        v = rng.normal(size=(T + 1, B)).astype(np.float32)
        r = (rng.random((T, B)) < 0.1).astype(np.float32)
        disc = ((rng.random((T, B)) > 0.05) * 0.99).astype(np.float32)
        rho = np.exp(rng.normal(0, 0.5, size=(T, B))).astype(np.float32)
        ins = [np.ascontiguousarray(x) for x in (v[:-1], v[1:], r, disc, rho)]
        bufs = [L.DevBuf(ctx, x) for x in ins]
        outs = [L.DevBuf(ctx, nbytes=T * B * 4, dtype=np.float32, shape=(T, B)) for _ in range(3)]
        L._chk(ctx.lib.cbm_vtrace(ctx.h, *[L._p(b.ptr) for b in bufs], T, B, *[L._p(o.ptr) for o in outs]))
        want = oracle.vtrace(*ins)
        for o, w in zip(outs, want):
            assert (bits(o.download()) == bits(w)).all(), (T, B)
        for b in bufs + outs:
            b.free()
    T, B = 16, 5
    v = rng.normal(size=(T + 1, B)).astype(np.float32)
    r = rng.random((T, B)).astype(np.float32)
    d = (rng.random((T + 1, B)) < 0.2).astype(np.uint8)
    disc = ((1 - d[1:]) * np.float32(0.99)).astype(np.float32)
    ins = [np.ascontiguousarray(x) for x in (v[:-1], v[1:], r, disc, np.ones((T, B), np.float32))]
    bufs = [L.DevBuf(ctx, x) for x in ins]
    outs = [L.DevBuf(ctx, nbytes=T * B * 4, dtype=np.float32, shape=(T, B)) for _ in range(3)]
    L._chk(ctx.lib.cbm_vtrace(ctx.h, *[L._p(b.ptr) for b in bufs], T, B, *[L._p(o.ptr) for o in outs]))
    adv, _ = oracle.gae(r, v[:-1], d[:-1], v[-1], d[-1], gamma=0.99, gae_lambda=1.0)
    np.testing.assert_allclose(outs[0].download(), adv, rtol=1e-5, atol=1e-5)


def test_async_gae_and_minibatch_advnorm(ctx, oracle):
    """legacy --async-batch-size: env-id-indexed returns (naturecnn:232-262, 467-531) bit-exact (same serial recursion per env), the
    per-minibatch advantage normalisation (naturecnn:540-541) within 1e-5.  R > 1024 exercises the kernel's row chunking."""
    from test_oracle_async import make_async_rollout
    for (R, B, NE, seed) in [(60, 4, 12, 1), (384, 20, 60, 2), (33, 5, 5, 3), (2560, 6, 120, 4), (1025, 3, 7, 5), (7, 1, 3, 6)]:
        env_ids, r, v, d = make_async_rollout(R, B, NE, seed)
        bufs = [L.DevBuf(ctx, x) for x in (env_ids, r, v, d)]
        dA = L.DevBuf(ctx, nbytes=R * B * 4, dtype=np.float32, shape=(R, B))
        dT = L.DevBuf(ctx, nbytes=R * B * 4, dtype=np.float32, shape=(R, B))
        L._chk(ctx.lib.cbm_gae_async(ctx.h, *[L._p(b.ptr) for b in bufs], R, B, NE, L._p(dA.ptr), L._p(dT.ptr)))
        adv_o, tgt_o = oracle.gae_async(env_ids, r, v, d, NE)
        assert (bits(dA.download()) == bits(adv_o)).all() and (bits(dT.download()) == bits(tgt_o)).all(), (R, B, NE)
        for b in bufs + [dT]:
            b.free()
        n = R * B
        perm = np.random.default_rng(seed).permutation(n).astype(np.int32)
        dI = L.DevBuf(ctx, perm)
        dN = L.DevBuf(ctx, np.zeros(n, np.float32))
        half = n // 2
        for lo, hi in ((0, half), (half, n)):   # two "minibatches" partition the permutation
            L._chk(ctx.lib.cbm_mb_advnorm(ctx.h, L._p(dA.ptr), L._p(dI.ptr + 4 * lo), hi - lo, L._p(dN.ptr)))
        got = dN.download()
        flat = adv_o.reshape(-1)
        for lo, hi in ((0, half), (half, n)):
            if hi - lo > 1 and flat[perm[lo:hi]].std() > 0:
                np.testing.assert_allclose(got[perm[lo:hi]], oracle.mb_advnorm(flat[perm[lo:hi]]), rtol=0, atol=2e-5)
        for b in (dA, dI, dN):
            b.free()


def test_ppo_loss_and_grads(ctx, oracle):
    rng = np.random.default_rng(7)
    N = 64
    P = make_params(A, 21)
    obs = make_frames(100, 22)
    idx = rng.permutation(100)[:N].astype(np.int32)
    actions = rng.integers(0, A, N).astype(np.int32)
    old_lp = (-np.log(A) + 0.2 * rng.normal(size=N)).astype(np.float32)
    adv = rng.normal(size=N).astype(np.float32)
    tgt = rng.normal(size=N).astype(np.float32)
    d = [L.DevBuf(ctx, x) for x in (P, obs, idx, actions, old_lp, adv, tgt)]
    dS = L.DevBuf(ctx, nbytes=32, dtype=np.float32)
    dG = L.DevBuf(ctx, nbytes=P.size * 4, dtype=np.float32)
    dLg = L.DevBuf(ctx, nbytes=N * A * 4, dtype=np.float32, shape=(N, A))
    dV = L.DevBuf(ctx, nbytes=N * 4, dtype=np.float32)
    L._chk(ctx.lib.cbm_ppo_loss_grad(ctx.h, L._p(d[0].ptr), L._p(d[1].ptr), L._p(d[2].ptr), N, L._p(d[3].ptr), L._p(d[4].ptr),
                                     L._p(d[5].ptr), L._p(d[6].ptr), L._p(dS.ptr), L._p(dG.ptr), L._p(dLg.ptr), L._p(dV.ptr)))
    stats_o, grads_o, logits_o, value_o = oracle.ppo_loss_grad(P, A, obs, idx, actions, old_lp, adv, tgt)
    assert (bits(dLg.download()) == bits(logits_o)).all()
    np.testing.assert_allclose(dS.download()[:5], stats_o, rtol=1e-5, atol=1e-6)
    g = dG.download()
    for name, (o, shp) in oracle.nature_layout(A).items():
        n = int(np.prod(shp))
        ref = grads_o[o:o + n]
        assert np.abs(g[o:o + n] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-7), name


def test_impala_loss_and_grads(ictx, oracle):
    rng = np.random.default_rng(8)
    T1, Bm = 8, 4
    N = T1 * Bm
    P = make_params(A, 31)
    obs = make_frames(N, 32)
    mu = rng.normal(0, 0.3, size=(T1, Bm, A)).astype(np.float32)
    actions = rng.integers(0, A, (T1, Bm)).astype(np.int32)
    rewards = (rng.random((T1, Bm)) < 0.3).astype(np.float32)
    dones = (rng.random((T1, Bm)) < 0.2).astype(np.uint8)
    first = (rng.random((T1, Bm)) < 0.2).astype(np.uint8)
    d = [L.DevBuf(ictx, x) for x in (P, obs, mu, actions, rewards, dones, first)]
    dS = L.DevBuf(ictx, nbytes=32, dtype=np.float32)
    dG = L.DevBuf(ictx, nbytes=P.size * 4, dtype=np.float32)
    L._chk(ictx.lib.cbm_impala_loss_grad(ictx.h, L._p(d[0].ptr), L._p(d[1].ptr), None, T1, Bm, L._p(d[2].ptr), L._p(d[3].ptr),
                                         L._p(d[4].ptr), L._p(d[5].ptr), L._p(d[6].ptr), L._p(dS.ptr), L._p(dG.ptr)))
    stats_o, grads_o = oracle.impala_loss_grad(P, A, obs, None, T1, Bm, mu, actions, rewards, dones, first)
    np.testing.assert_allclose(dS.download()[:4], stats_o, rtol=1e-5, atol=1e-5)
    g = dG.download()
    for name, (o, shp) in oracle.nature_layout(A).items():
        n = int(np.prod(shp))
        ref = grads_o[o:o + n]
        assert np.abs(g[o:o + n] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-7), name


def test_optimizers(ctx, oracle):
    rng = np.random.default_rng(9)
    n = 1693875
    p0 = rng.normal(size=n).astype(np.float32)
    for scale in (1e-5, 1.0):
        p, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
        dp, dm, dv = L.DevBuf(ctx, p), L.DevBuf(ctx, m), L.DevBuf(ctx, v)
        for step in (1, 2, 3):
            g = (scale * rng.normal(size=n)).astype(np.float32)
            bc1 = float(np.float32(1) - np.power(np.float32(0.9), np.float32(step)))
            bc2 = float(np.float32(1) - np.power(np.float32(0.999), np.float32(step)))
            oracle.adam_step(p, g, m, v, 0.5, 2.5e-4, bc1=bc1, bc2=bc2)
            dg = L.DevBuf(ctx, g)
            L._chk(ctx.lib.cbm_adam_step(ctx.h, L._p(dp.ptr), L._p(dg.ptr), L._p(dm.ptr), L._p(dv.ptr), L.C.c_int64(n), L.C.c_float(0.5),
                                         L.C.c_float(2.5e-4), L.C.c_float(bc1), L.C.c_float(bc2), L.C.c_float(1.0)))
            dg.free()
        np.testing.assert_allclose(dp.download(), p, rtol=0, atol=1e-6)
        np.testing.assert_allclose(dm.download(), m, rtol=1e-5, atol=1e-9)
    p, nu = p0.copy(), np.zeros(n, np.float32)
    dp, dn = L.DevBuf(ctx, p), L.DevBuf(ctx, nu)
    for step in range(2):
        g = (100 * rng.normal(size=n)).astype(np.float32)     # above the clip threshold 40
        oracle.rmsprop_step(p, g, nu, 40.0, 6e-4)
        dg = L.DevBuf(ctx, g)
        L._chk(ctx.lib.cbm_rmsprop_step(ctx.h, L._p(dp.ptr), L._p(dg.ptr), L._p(dn.ptr), L.C.c_int64(n), L.C.c_float(40.0),
                                        L.C.c_float(6e-4), L.C.c_float(1.0)))
        dg.free()
    np.testing.assert_allclose(dp.download(), p, rtol=0, atol=1e-6)


def test_bf16_forward_extension_close_to_fp32_oracle(oracle):
    # BASELINE configs[2] / SURVEY §8d: bf16 forward is a build-only extension; tolerance vs the fp32 oracle 2e-2 on logits
    cfg = L.default_config(L.ALGO_PPO)
    cfg.forward_bf16 = 1
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 64, 1, 48   # learner workspace for 768-frame minibatches
    ctx = L.Context(cfg)
    try:
        A = 18
        P = make_params(A, 5)
        obs = make_frames(700, 9)           # > 512 frames: the learner-size code path; 12 frames: the actor-size path (split-K dense)
        for B, ks in ((700, 1), (12, 14)):
            dP, dO = L.DevBuf(ctx, P), L.DevBuf(ctx, obs[:B])
            dL = L.DevBuf(ctx, nbytes=B * A * 4, dtype=np.float32, shape=(B, A))
            dV = L.DevBuf(ctx, nbytes=B * 4, dtype=np.float32, shape=(B,))
            L._chk(ctx.lib.cbm_forward(ctx.h, L._p(dP.ptr), L._p(dO.ptr), None, B, ks, L._p(dL.ptr), L._p(dV.ptr)))
            lo, vo = oracle.nature_forward(P, A, obs[:B], ksplit=ks)
            lg, vg = dL.download(), dV.download()
            assert np.isfinite(lg).all()
            assert np.abs(lg - lo).max() <= 2e-2 * max(1.0, np.abs(lo).max()), np.abs(lg - lo).max()
            assert np.abs(vg - vo).max() <= 2e-2 * max(1.0, np.abs(vo).max())
            assert np.abs(lg - lo).max() > 0      # it really is a different arithmetic (not silently the fp32 path)
    finally:
        ctx.close()


@pytest.mark.parametrize("na", [4, 6, 9, 28])
def test_other_action_set_sizes(oracle, na):
    # envpool games with reduced action sets (Breakout minimal = 4, Pong = 6, ...; SURVEY 8a: "A=18 ... or 4") and the ABI maximum 28:
    # forward / sampling bit-exact, one PPO update within 1e-5 of the oracle-backed engine
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    import cleanba_amd.model as M
    import cleanba_amd.prng as prng
    E, T = 8, 8
    cfg = L.default_config(L.ALGO_PPO)
    cfg.num_actions, cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.update_epochs = na, E, 1, T, 2
    ctx = L.Context(cfg)
    try:
        key = prng.prng_key(3)
        key, nk, ak, ck = prng.split(key, 4)
        params = M.init_nature_params(na, nk, ak, ck)
        ctx.set_params(params)
        ctx.actor_set_key(0, key)
        ctx.actor_env_reset_device(0, 9)
        ctx.actor_begin_rollout(0, False)
        ctx.actor_rollout_device(0, T)
        ctx.actor_commit(0)
        ctx.learner_wait()
        obs = ctx.read("obs", np.uint8).reshape(T + 1, E, 4, 84, 84)
        actions = ctx.read("actions", np.int32).reshape(T + 1, E)
        logprobs = ctx.read("logprobs", np.float32).reshape(T + 1, E)
        k = key.copy()
        for t in range(T):   # replay the policy part of the rollout
            logits, value = oracle.nature_forward(params, na, obs[t], ksplit=cfg.actor_dense_ksplit)
            a, lp, k = oracle.sample_actions(logits, k)
            assert (a == actions[t]).all() and (bits(lp) == bits(logprobs[t])).all() and a.max() < na
        ora = OracleEngine(cfg)
        ora.set_params(params)
        R = ora.ring[0]
        R["obs"][:] = obs
        for name, dt in (("actions", np.int32), ("logprobs", np.float32), ("values", np.float32), ("rewards", np.float32), ("dones", np.uint8)):
            R[name][:] = ctx.read(name, dt).reshape(T + 1, E)
        ora.committed = [1]
        n_opt = 8
        bc = [M.adam_bias_corrections(i + 1) for i in range(n_opt)]
        lrs, b1, b2 = np.full(n_opt, 1e-3, np.float32), np.array([b[0] for b in bc], np.float32), np.array([b[1] for b in bc], np.float32)
        _, s_g = ctx.learner_update(key, lrs, b1, b2)
        _, s_o = ora.learner_update(key, lrs, b1, b2)
        np.testing.assert_allclose(s_g, s_o, rtol=2e-4, atol=2e-5)
        p_g, p_o = ctx.get_params(), ora.get_params()
        assert np.abs(p_g - p_o).max() <= 1e-5 * max(1.0, np.abs(p_o).max())
    finally:
        ctx.close()


def test_ppo_grads_at_ragged_batch_sizes(oracle):
    """Backward parity at sizes that do not fill the position-major conv3 dgrad's 128-frame tiles (130 = one full tile + 2 frames, 7 = a
    sliver), with a gather index — the masked rows of the last frame tile, the K-skip table and the bit-mask epilogues all take their edge paths."""
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 68, 1, 8      # MB = 136 frames of workspace
    c = L.Context(cfg)
    rng = np.random.default_rng(70)
    P = make_params(A, 71)
    pool = make_frames(200, 72)
    dP, dO = L.DevBuf(c, P), L.DevBuf(c, pool)
    for N in (130, 7):
        idx = rng.permutation(200)[:N].astype(np.int32)
        actions = rng.integers(0, A, N).astype(np.int32)
        old_lp = (-np.log(A) + 0.2 * rng.normal(size=N)).astype(np.float32)
        adv = rng.normal(size=N).astype(np.float32)
        tgt = rng.normal(size=N).astype(np.float32)
        d = [L.DevBuf(c, x) for x in (idx, actions, old_lp, adv, tgt)]
        dS = L.DevBuf(c, nbytes=32, dtype=np.float32)
        dG = L.DevBuf(c, nbytes=P.size * 4, dtype=np.float32)
        L._chk(c.lib.cbm_ppo_loss_grad(c.h, L._p(dP.ptr), L._p(dO.ptr), L._p(d[0].ptr), N, L._p(d[1].ptr), L._p(d[2].ptr), L._p(d[3].ptr),
                                       L._p(d[4].ptr), L._p(dS.ptr), L._p(dG.ptr), None, None))
        stats_o, grads_o, _, _ = oracle.ppo_loss_grad(P, A, pool, idx, actions, old_lp, adv, tgt)
        np.testing.assert_allclose(dS.download()[:5], stats_o, rtol=1e-5, atol=1e-6)
        g = dG.download()
        for name, (o, shp) in oracle.nature_layout(A).items():
            n = int(np.prod(shp))
            ref = grads_o[o:o + n]
            assert np.abs(g[o:o + n] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-7), (N, name)
    c.close()


@pytest.mark.parametrize("ns", [2, 3])
def test_backward_split_bf16_meets_the_fp32_bar(oracle, ns):
    """cbm_config.backward_split (build-only extension): the backward GEMMs on bf16 MFMA with every fp32 operand split exactly into 2 / 3
    bf16 terms.  The bar is the SAME as for the fp32 MFMA path — gradients within 1e-5 of the oracle per tensor, losses untouched, forward
    logits still bit-exact — and the option must not silently be the fp32 path."""
    out = {}
    rng = np.random.default_rng(80)
    N = 130
    P = make_params(A, 81)
    pool = make_frames(160, 82)
    idx = rng.permutation(160)[:N].astype(np.int32)
    actions = rng.integers(0, A, N).astype(np.int32)
    old_lp = (-np.log(A) + 0.2 * rng.normal(size=N)).astype(np.float32)
    adv = rng.normal(size=N).astype(np.float32)
    tgt = rng.normal(size=N).astype(np.float32)
    stats_o, grads_o, logits_o, _ = oracle.ppo_loss_grad(P, A, pool, idx, actions, old_lp, adv, tgt)
    for split in (0, ns):
        cfg = L.default_config(L.ALGO_PPO)
        cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 68, 1, 8
        cfg.backward_split = split
        c = L.Context(cfg)
        d = [L.DevBuf(c, x) for x in (P, pool, idx, actions, old_lp, adv, tgt)]
        dS = L.DevBuf(c, nbytes=32, dtype=np.float32)
        dG = L.DevBuf(c, nbytes=P.size * 4, dtype=np.float32)
        dLg = L.DevBuf(c, nbytes=N * A * 4, dtype=np.float32, shape=(N, A))
        L._chk(c.lib.cbm_ppo_loss_grad(c.h, L._p(d[0].ptr), L._p(d[1].ptr), L._p(d[2].ptr), N, L._p(d[3].ptr), L._p(d[4].ptr), L._p(d[5].ptr),
                                       L._p(d[6].ptr), L._p(dS.ptr), L._p(dG.ptr), L._p(dLg.ptr), None))
        assert (bits(dLg.download()) == bits(logits_o)).all()
        np.testing.assert_allclose(dS.download()[:5], stats_o, rtol=1e-5, atol=1e-6)
        g = dG.download()
        for name, (o, shp) in oracle.nature_layout(A).items():
            n = int(np.prod(shp))
            ref = grads_o[o:o + n]
            assert np.abs(g[o:o + n] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-7), (split, name)
        out[split] = g
        c.close()
    assert not np.array_equal(out[0], out[ns])


def test_backward_split_impala_grads(oracle):
    """The split-bf16 backward under the IMPALA loss (same backward kernels, V-trace loss head): gradients within 1e-5 of the oracle per tensor."""
    rng = np.random.default_rng(90)
    T1, Bm = 9, 15          # 135 frames: a full 128-frame tile plus a sliver
    N = T1 * Bm
    P = make_params(A, 91)
    obs = make_frames(N, 92)
    mu = rng.normal(0, 0.3, size=(T1, Bm, A)).astype(np.float32)
    actions = rng.integers(0, A, (T1, Bm)).astype(np.int32)
    rewards = (rng.random((T1, Bm)) < 0.3).astype(np.float32)
    dones = (rng.random((T1, Bm)) < 0.2).astype(np.uint8)
    first = (rng.random((T1, Bm)) < 0.2).astype(np.uint8)
    stats_o, grads_o = oracle.impala_loss_grad(P, A, obs, None, T1, Bm, mu, actions, rewards, dones, first)
    cfg = L.default_config(L.ALGO_IMPALA)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 60, 1, T1 - 1
    cfg.backward_split = 2
    c = L.Context(cfg)
    d = [L.DevBuf(c, x) for x in (P, obs, mu, actions, rewards, dones, first)]
    dS = L.DevBuf(c, nbytes=32, dtype=np.float32)
    dG = L.DevBuf(c, nbytes=P.size * 4, dtype=np.float32)
    L._chk(c.lib.cbm_impala_loss_grad(c.h, L._p(d[0].ptr), L._p(d[1].ptr), None, T1, Bm, L._p(d[2].ptr), L._p(d[3].ptr), L._p(d[4].ptr), L._p(d[5].ptr),
                                      L._p(d[6].ptr), L._p(dS.ptr), L._p(dG.ptr)))
    np.testing.assert_allclose(dS.download()[:4], stats_o, rtol=1e-5, atol=1e-5)
    g = dG.download()
    for name, (o, shp) in oracle.nature_layout(A).items():
        n = int(np.prod(shp))
        ref = grads_o[o:o + n]
        assert np.abs(g[o:o + n] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-7), name
    c.close()
