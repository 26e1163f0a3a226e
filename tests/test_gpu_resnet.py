"""GPU parity for the IMPALA-ResNet torso (ppo:149-189): forward bit-exact, PPO loss/grads within 1e-5 of the oracle,
and one device-env rollout step replayed with the oracle (bit-exact actions)."""
import numpy as np
import pytest

import cleanba_amd.lib as L
from helpers import make_frames
from test_oracle_resnet import make_resnet_params

pytestmark = pytest.mark.gpu
A = 18


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def rctx():
    cfg = L.default_config(L.ALGO_PPO)
    cfg.network = L.NET_IMPALA_RESNET
    cfg.actor_dense_ksplit = 11
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 8, 1, 8
    c = L.Context(cfg)
    yield c
    c.close()


def test_resnet_forward_bit_exact(rctx, oracle):
    assert L.param_count(L.NET_IMPALA_RESNET, A) == oracle.resnet_param_count(A) == 1094115
    P = make_resnet_params(oracle, 4)
    obs = make_frames(12, 5)
    dP, dO = L.DevBuf(rctx, P), L.DevBuf(rctx, obs)
    for idx, ks in ((None, 1), (None, 11), ([3, 11, 0, 7, 7], 1)):
        B = len(idx) if idx is not None else 12
        dI = L.DevBuf(rctx, np.asarray(idx, np.int32)) if idx is not None else None
        dL = L.DevBuf(rctx, nbytes=B * A * 4, dtype=np.float32, shape=(B, A))
        dV = L.DevBuf(rctx, nbytes=B * 4, dtype=np.float32, shape=(B,))
        L._chk(rctx.lib.cbm_forward(rctx.h, L._p(dP.ptr), L._p(dO.ptr), L._p(dI.ptr if dI else None), B, ks, L._p(dL.ptr), L._p(dV.ptr)))
        lo, vo = oracle.resnet_forward(P, A, obs, idx=idx, ksplit=ks)
        lg, vg = dL.download(), dV.download()
        np.testing.assert_allclose(lg, lo, rtol=0, atol=1e-5 * max(1.0, np.abs(lo).max()))
        assert (bits(lg) == bits(lo)).all() and (bits(vg) == bits(vo)).all()


def test_resnet_actor_batch_forward_bit_exact(oracle):
    """120 frames (the actor's batch): the 11x11 layers run on the row-ring kernel (rnconv_rw.h) from 64 frames up, the rest on the slab kernel."""
    cfg = L.default_config(L.ALGO_PPO)
    cfg.network = L.NET_IMPALA_RESNET
    cfg.actor_dense_ksplit = 11
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 120, 1, 8
    ctx = L.Context(cfg)
    try:
        P = make_resnet_params(oracle, 11)
        for B in (120, 77):
            obs = make_frames(B, 13)
            dP, dO = L.DevBuf(ctx, P), L.DevBuf(ctx, obs)
            dL = L.DevBuf(ctx, nbytes=B * A * 4, dtype=np.float32, shape=(B, A))
            dV = L.DevBuf(ctx, nbytes=B * 4, dtype=np.float32, shape=(B,))
            L._chk(ctx.lib.cbm_forward(ctx.h, L._p(dP.ptr), L._p(dO.ptr), None, B, 11, L._p(dL.ptr), L._p(dV.ptr)))
            lo, vo = oracle.resnet_forward(P, A, obs, ksplit=11)
            assert (bits(dL.download()) == bits(lo)).all() and (bits(dV.download()) == bits(vo)).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("N", [16, 5, 1])   # 5 / 1: partial frame pairs in the 11x11 layers, fewer strips than persistent wgrad blocks
def test_resnet_ppo_loss_and_grads(rctx, oracle, N):
    rng = np.random.default_rng(6)
    P = make_resnet_params(oracle, 7)
    obs = make_frames(24, 8)
    idx = rng.permutation(24)[:N].astype(np.int32)
    actions = rng.integers(0, A, N).astype(np.int32)
    old_lp = (-np.log(A) + 0.2 * rng.normal(size=N)).astype(np.float32)
    adv = rng.normal(size=N).astype(np.float32)
    tgt = rng.normal(size=N).astype(np.float32)
    d = [L.DevBuf(rctx, x) for x in (P, obs, idx, actions, old_lp, adv, tgt)]
    dS = L.DevBuf(rctx, nbytes=32, dtype=np.float32)
    dG = L.DevBuf(rctx, nbytes=P.size * 4, dtype=np.float32)
    L._chk(rctx.lib.cbm_ppo_loss_grad(rctx.h, L._p(d[0].ptr), L._p(d[1].ptr), L._p(d[2].ptr), N, L._p(d[3].ptr), L._p(d[4].ptr), L._p(d[5].ptr),
                                      L._p(d[6].ptr), L._p(dS.ptr), L._p(dG.ptr), None, None))
    logits, value, acts = oracle.resnet_forward(P, A, obs, idx=idx, save_acts=True)
    stats, dlog, dval = oracle.ppo_loss_head(logits, value, actions, old_lp, adv, tgt)
    grads_o = oracle.resnet_backward(P, A, obs, idx, acts, dlog, dval)
    np.testing.assert_allclose(dS.download()[:5], stats, rtol=1e-5, atol=1e-6)
    g = dG.download()
    for name, (o, shp) in oracle.resnet_layout(A).items():
        n = int(np.prod(shp))
        ref = grads_o[o:o + n]
        assert np.abs(g[o:o + n] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-7), (name, np.abs(g[o:o + n] - ref).max(), np.abs(ref).max())


def test_resnet_device_rollout_replayed_by_the_oracle(oracle):
    """A device-env rollout on the IMPALA-ResNet (the actor step ends in actor_tail_rows_kernel<256>: split-K reduce + heads + sampling + env step in
    one launch) replayed with the host env twin + the oracle policy: frames, actions, rewards, dones, log-probs and values bit for bit (ppo:245-261, 308-353)."""
    import cleanba_amd.prng as prng
    E, T = 8, 6
    cfg = L.default_config(L.ALGO_PPO)
    cfg.network = L.NET_IMPALA_RESNET
    cfg.actor_dense_ksplit = 11
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, 1, T
    ctx = L.Context(cfg)
    try:
        params = make_resnet_params(oracle, 21)
        key = prng.prng_key(3)
        ctx.set_params(params)
        ctx.actor_set_key(0, key)
        ctx.actor_env_reset_device(0, 9)
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
        st, o = L.synth_env_reset_host(9, E)
        k = key.copy()
        for t in range(T):
            assert (o == obs[t]).all(), f"frames differ at t={t}"
            logits, value = oracle.resnet_forward(params, A, o, ksplit=cfg.actor_dense_ksplit)
            a, lp, k = oracle.sample_actions(logits, k)
            assert (a == actions[t]).all(), f"sampled actions differ at t={t}"
            assert (bits(lp) == bits(logprobs[t])).all() and (bits(value) == bits(values[t])).all()
            r, d, _, _ = L.synth_env_step_host(9, st, o, a)
            assert (r == rewards[t]).all() and (d == dones[t + 1]).all()
        assert (o == obs[T]).all() and (ctx.actor_get_key(0) == k).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("which", ["reg", "lds"])
def test_resnet_first_conv_pool_kernels_agree_with_the_oracle(rctx, oracle, which, monkeypatch):
    """The first conv + max_pool has two kernels: strips with the conv rows in LDS (small batches) and one wave per frame pooling on the accumulators
    (learner batches, B >= 1024).  CBM_RN_POOL0 forces either at any batch: logits / values bit for bit, gradients (through the arg-max bytes and the
    relu mask bits the kernel writes) within the usual bar — at 12 / 5 / 16 frames, with and without a gather index."""
    monkeypatch.setenv("CBM_RN_POOL0", which)
    test_resnet_forward_bit_exact(rctx, oracle)
    for N in (16, 5):
        test_resnet_ppo_loss_and_grads(rctx, oracle, N)
