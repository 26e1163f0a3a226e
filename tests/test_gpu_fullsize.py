"""BASELINE.json's full sizes (configs[1] PPO / configs[2] IMPALA: E=120, T=128, minibatches of 3840 / 3870 frames) checked through
size-independent properties — the CPU oracle needs ~1.5 ms per frame and cannot replay them in test time:
  * the learner-size kernels (frame-resident conv1, 128x64 tiles, DMA-staged dense) against the actor-size kernels, which the other
    GPU tests pin to the oracle bit for bit: same frames, same forward chain -> identical bits;
  * backward linearity: with ratio = 1 and no value / entropy terms the PPO gradient is linear in the advantages;
  * a full rollout + update is a deterministic function of (seed, env seed)."""
import numpy as np
import pytest

import cleanba_amd.lib as L
import cleanba_amd.model as M
import cleanba_amd.prng as prng
from helpers import make_frames, make_params

pytestmark = pytest.mark.gpu
A, E, T = 18, 120, 128
MB = E * T // 4


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def big():
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, 1, T
    cfg.vf_coef, cfg.ent_coef = 0.0, 0.0
    cfg.conv1_fp32_chain = 3        # learner-size conv1 on the fp32 chain kernels: this file's bitwise identities (learner-size == actor-size kernels, split ==
    c = L.Context(cfg)              # whole) are properties of ONE summation order; the default exact-product conv1 is held to the oracle in test_gpu_conv1_exact.py
    yield c
    c.close()


def _forward(ctx, dP, dO, B, off_frames=0):
    dL = L.DevBuf(ctx, nbytes=B * A * 4, dtype=np.float32, shape=(B, A))
    dV = L.DevBuf(ctx, nbytes=B * 4, dtype=np.float32, shape=(B,))
    L._chk(ctx.lib.cbm_forward(ctx.h, L._p(dP.ptr), L._p(dO.ptr + off_frames * L.FRAME), None, B, 1, L._p(dL.ptr), L._p(dV.ptr)))
    out = dL.download(), dV.download()
    dL.free(); dV.free()
    return out


def test_learner_size_forward_equals_actor_size_forward(big):
    obs = make_frames(MB, 41)
    P = make_params(A, 42)
    dP, dO = L.DevBuf(big, P), L.DevBuf(big, obs)
    lg, vg = _forward(big, dP, dO, MB)                       # one 3840-frame launch chain (big-path kernels)
    for lo in range(0, MB, 480):                             # the same frames, 480 at a time (small-path kernels)
        ls, vs = _forward(big, dP, dO, 480, off_frames=lo)
        assert (bits(ls) == bits(lg[lo:lo + 480])).all() and (bits(vs) == bits(vg[lo:lo + 480])).all(), lo
    assert np.isfinite(lg).all() and np.abs(lg).max() > 0


def _pg_grads(ctx, dP, dO, idx, actions, logp, adv):
    n = len(idx)
    tgt = np.zeros(n, np.float32)
    d = [L.DevBuf(ctx, x) for x in (idx, actions, logp, adv, tgt)]
    dS = L.DevBuf(ctx, nbytes=32, dtype=np.float32)
    dG = L.DevBuf(ctx, nbytes=ctx.P * 4, dtype=np.float32)
    dL = L.DevBuf(ctx, nbytes=n * A * 4, dtype=np.float32, shape=(n, A))
    L._chk(ctx.lib.cbm_ppo_loss_grad(ctx.h, L._p(dP.ptr), L._p(dO.ptr), L._p(d[0].ptr), n, L._p(d[1].ptr), L._p(d[2].ptr), L._p(d[3].ptr),
                                      L._p(d[4].ptr), L._p(dS.ptr), L._p(dG.ptr), L._p(dL.ptr), None))
    out = dG.download().astype(np.float64), dL.download()
    for b in d + [dS, dG, dL]:
        b.free()
    return out


def test_full_minibatch_gradient_is_linear_in_the_advantages(big):
    rng = np.random.default_rng(43)
    obs = make_frames(MB, 44)
    P = make_params(A, 45)
    dP, dO = L.DevBuf(big, P), L.DevBuf(big, obs)
    idx = rng.permutation(MB).astype(np.int32)
    actions = rng.integers(0, A, MB).astype(np.int32)
    zero = np.zeros(MB, np.float32)
    _, logits = _pg_grads(big, dP, dO, idx, actions, zero, zero)
    z = logits.astype(np.float64)
    z -= z.max(1, keepdims=True)
    logp = (z - np.log(np.exp(z).sum(1, keepdims=True)))[np.arange(MB), actions].astype(np.float32)   # old logprob = new -> ratio 1
    a1 = rng.normal(size=MB).astype(np.float32)
    a2 = rng.normal(size=MB).astype(np.float32)
    g1, _ = _pg_grads(big, dP, dO, idx, actions, logp, a1)
    g2, _ = _pg_grads(big, dP, dO, idx, actions, logp, a2)
    g12, _ = _pg_grads(big, dP, dO, idx, actions, logp, (a1 + a2).astype(np.float32))
    scale = max(np.abs(g12).max(), 1e-12)
    assert np.abs(g1).max() > 1e-6 * scale and np.abs(g12 - (g1 + g2)).max() <= 2e-5 * scale


@pytest.mark.parametrize("algo", [L.ALGO_PPO, L.ALGO_IMPALA])
def test_full_size_step_is_deterministic(algo):
    def run(env_seed):
        cfg = L.default_config(algo)
        cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, 1, T
        ctx = L.Context(cfg)
        key = prng.prng_key(1)
        key, nk, ak, ck = prng.split(key, 4)
        ctx.set_params(M.init_nature_params(A, nk, ak, ck))
        ctx.actor_set_key(0, key)
        ctx.actor_env_reset_device(0, env_seed)
        ctx.actor_begin_rollout(0, False)
        ctx.actor_rollout_device(0, T + (0 if algo == L.ALGO_PPO else 1))
        ctx.actor_commit(0)
        ctx.learner_wait()
        n_opt = 16 if algo == L.ALGO_PPO else 4
        bc = [M.adam_bias_corrections(i + 1) for i in range(n_opt)]
        _, stats = ctx.learner_update(key, np.full(n_opt, 2.5e-4, np.float32), np.array([b[0] for b in bc], np.float32),
                                      np.array([b[1] for b in bc], np.float32))
        p = ctx.get_params()
        actions = ctx.read("actions", np.int32)
        ctx.close()
        return p, stats, actions
    p1, s1, a1 = run(7)
    p2, s2, a2 = run(7)
    p3, _, a3 = run(8)
    assert np.isfinite(p1).all() and np.isfinite(s1).all()
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2) and np.array_equal(a1, a2)
    assert not np.array_equal(a1, a3) and not np.array_equal(p1, p3)


def test_resnet_large_batch_properties():
    # the reference's default torso at a 1024-frame minibatch: chunked forward == one-shot forward (bits), gradient linear in advantages
    from test_oracle_resnet import make_resnet_params
    import oracle
    cfg = L.default_config(L.ALGO_PPO)
    cfg.network, cfg.actor_dense_ksplit = L.NET_IMPALA_RESNET, 11
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 64, 1, 64
    cfg.vf_coef, cfg.ent_coef = 0.0, 0.0
    ctx = L.Context(cfg)
    try:
        n = 1024
        rng = np.random.default_rng(51)
        obs = make_frames(n, 52)
        P = make_resnet_params(oracle, 53)
        dP, dO = L.DevBuf(ctx, P), L.DevBuf(ctx, obs)
        lg, vg = _forward(ctx, dP, dO, n)
        for lo in (0, 256, 768):
            ls, vs = _forward(ctx, dP, dO, 256 if lo < 768 else 255, off_frames=lo)
            m = len(vs)
            assert (bits(ls) == bits(lg[lo:lo + m])).all() and (bits(vs) == bits(vg[lo:lo + m])).all()
        idx = rng.permutation(n).astype(np.int32)
        actions = rng.integers(0, A, n).astype(np.int32)
        zero = np.zeros(n, np.float32)
        _, logits = _pg_grads(ctx, dP, dO, idx, actions, zero, zero)
        z = logits.astype(np.float64)
        z -= z.max(1, keepdims=True)
        logp = (z - np.log(np.exp(z).sum(1, keepdims=True)))[np.arange(n), actions].astype(np.float32)
        a1, a2 = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
        g1, _ = _pg_grads(ctx, dP, dO, idx, actions, logp, a1)
        g2, _ = _pg_grads(ctx, dP, dO, idx, actions, logp, a2)
        g12, _ = _pg_grads(ctx, dP, dO, idx, actions, logp, (a1 + a2).astype(np.float32))
        scale = max(np.abs(g12).max(), 1e-12)
        assert np.abs(g1).max() > 1e-6 * scale and np.abs(g12 - (g1 + g2)).max() <= 2e-5 * scale
    finally:
        ctx.close()


def test_relu_bit_masks_equal_the_activations(big):
    """The forward epilogues emit the ReLU masks as bits (ballot words collected with v_writelane — inline asm, so the hazard wait states
    are ours to keep); the dgrad epilogues trust them.  After a full-size forward every word must equal (activation > 0)."""
    P = make_params(A, 51)
    for rep, n in enumerate((MB, 1000, MB)):
        obs = make_frames(n, 52 + rep)
        dP, dO = L.DevBuf(big, P), L.DevBuf(big, obs)
        L._chk(big.lib.cbm_forward(big.h, L._p(dP.ptr), L._p(dO.ptr), None, n, 1 if n > 1024 else 14, None, None))
        for name, ch, rows in (("1", 32, n * 400), ("2", 64, n * 81), ("3", 64, n * 49)):
            act = big.read("lws_act" + name, np.float32).reshape(-1, ch)[:rows]
            m = big.read("lws_mask" + name, np.uint32).reshape(-1, ch // 32)[:rows]
            for w in range(ch // 32):
                bits = (act[:, 32 * w:32 * w + 32] > 0).astype(np.uint64)
                want = (bits << np.arange(32, dtype=np.uint64)).sum(1).astype(np.uint32)
                bad = np.nonzero(want != m[:, w])[0]
                assert bad.size == 0, (n, name, w, bad[:8])
        dP.free(); dO.free()


@pytest.mark.parametrize("epochs,E2,T2", [(4, E, T), (3, 16, 32), (1, 16, 32)])
def test_whole_update_call_equals_the_per_epoch_calls(epochs, E2, T2):
    """cbm_learner_update permutes every epoch of the update in one batch of launches (and lets the loss statistics ride in the reduction
    launch); the split C calls (prepare / epoch_begin / minibatch_grad / optimizer_step / finish, the form a data-parallel host drives)
    permute epoch by epoch.  Same rollout, same key -> the same key afterwards, the same statistics rows and the same parameters, bit for bit
    (ppo:599-615: one subkey per epoch)."""
    def run(split):
        cfg = L.default_config(L.ALGO_PPO)
        cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.update_epochs = E2, 1, T2, epochs
        ctx = L.Context(cfg)
        key = prng.prng_key(3)
        key, nk, ak, ck = prng.split(key, 4)
        ctx.set_params(M.init_nature_params(A, nk, ak, ck))
        ctx.actor_set_key(0, key)
        ctx.actor_env_reset_device(0, 5)
        ctx.actor_begin_rollout(0, False)
        ctx.actor_rollout_device(0, T2)
        ctx.actor_commit(0)
        ctx.learner_wait()
        nmb = cfg.num_minibatches
        n_opt = epochs * nmb
        lrs = np.full(n_opt, 2.5e-4, np.float32)
        bc = [M.adam_bias_corrections(i + 1) for i in range(n_opt)]
        b1, b2 = np.array([b[0] for b in bc], np.float32), np.array([b[1] for b in bc], np.float32)
        if not split:
            k, stats = ctx.learner_update(key, lrs, b1, b2)
        else:
            k = ctx.learner_prepare(key)
            i = 0
            for e in range(epochs):
                k = ctx.learner_epoch_begin(k)
                for mb in range(nmb):
                    ctx.learner_minibatch_grad(e, mb)
                    ctx.learner_optimizer_step(float(lrs[i]), float(b1[i]), float(b2[i]))
                    i += 1
            stats = ctx.learner_finish(n_opt)
        p = ctx.get_params()
        ctx.close()
        return np.asarray(k), np.asarray(stats), p
    k1, s1, p1 = run(False)
    k2, s2, p2 = run(True)
    assert np.isfinite(p1).all() and np.isfinite(s1).all()
    assert np.array_equal(k1, k2)
    assert np.array_equal(bits(s1), bits(s2))
    assert np.array_equal(bits(p1), bits(p2))
