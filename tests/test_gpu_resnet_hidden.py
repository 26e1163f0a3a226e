"""`--hiddens H` on the IMPALA-ResNet torso (Network.hiddens, ppo:94): the HIP path with a non-default hidden width against the oracle
with the same width — forward bit-exact, PPO gradients within 1e-5 — and the widths the torso is not built for are refused."""
import numpy as np
import pytest

import cleanba_amd.lib as L
import cleanba_amd.model as M
from helpers import make_frames
from test_oracle_resnet import make_resnet_params

pytestmark = pytest.mark.gpu
A = 18


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def _cfg(hid=None, channels=None, frames=16):
    cfg = L.default_config(L.ALGO_PPO)
    cfg.network = L.NET_IMPALA_RESNET
    cfg.actor_dense_ksplit = 11
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_minibatches = (frames + 7) // 8, 1, 8, 1   # learner workspace >= frames
    if hid is not None:
        cfg.num_hiddens = len(hid)
        for i, h in enumerate(hid):
            cfg.hiddens[i] = h
    if channels is not None:
        cfg.num_channels = len(channels)
        for i, c in enumerate(channels):
            cfg.channels[i] = c
    return cfg


@pytest.fixture
def hidden_oracle(oracle):
    yield oracle
    oracle.set_resnet_hidden(256)


@pytest.mark.parametrize("hid,N", [(128, 16), (512, 16), (64, 5), (448, 300)])
def test_resnet_hidden_width_parity(hidden_oracle, hid, N):
    oracle = hidden_oracle
    oracle.set_resnet_hidden(hid)
    ctx = L.Context(_cfg([hid], frames=max(N, 16)))
    try:
        assert ctx.P == L.param_count(L.NET_IMPALA_RESNET, A, hid) == oracle.resnet_param_count(A) == M.resnet_layout(A, hid)[1]
        rng = np.random.default_rng(hid)
        P = make_resnet_params(oracle, 9)
        assert P.size == ctx.P
        obs = make_frames(max(24, N), 8)
        idx = rng.permutation(obs.shape[0])[:N].astype(np.int32)
        actions = rng.integers(0, A, N).astype(np.int32)
        old_lp = (-np.log(A) + 0.2 * rng.normal(size=N)).astype(np.float32)
        adv = rng.normal(size=N).astype(np.float32)
        tgt = rng.normal(size=N).astype(np.float32)
        d = [L.DevBuf(ctx, x) for x in (P, obs, idx, actions, old_lp, adv, tgt)]
        for ks in (1, 11):
            dL = L.DevBuf(ctx, nbytes=N * A * 4, dtype=np.float32, shape=(N, A))
            dV = L.DevBuf(ctx, nbytes=N * 4, dtype=np.float32, shape=(N,))
            L._chk(ctx.lib.cbm_forward(ctx.h, L._p(d[0].ptr), L._p(d[1].ptr), L._p(d[2].ptr), N, ks, L._p(dL.ptr), L._p(dV.ptr)))
            lo, vo = oracle.resnet_forward(P, A, obs, idx=idx, ksplit=ks)
            assert (bits(dL.download()) == bits(lo)).all() and (bits(dV.download()) == bits(vo)).all()
        dS = L.DevBuf(ctx, nbytes=32, dtype=np.float32)
        dG = L.DevBuf(ctx, nbytes=P.size * 4, dtype=np.float32)
        L._chk(ctx.lib.cbm_ppo_loss_grad(ctx.h, L._p(d[0].ptr), L._p(d[1].ptr), L._p(d[2].ptr), N, L._p(d[3].ptr), L._p(d[4].ptr),
                                         L._p(d[5].ptr), L._p(d[6].ptr), L._p(dS.ptr), L._p(dG.ptr), None, None))
        logits, value, acts = oracle.resnet_forward(P, A, obs, idx=idx, save_acts=True)
        stats, dlog, dval = oracle.ppo_loss_head(logits, value, actions, old_lp, adv, tgt)
        grads_o = oracle.resnet_backward(P, A, obs, idx, acts, dlog, dval)
        np.testing.assert_allclose(dS.download()[:5], stats, rtol=1e-5, atol=1e-6)
        g = dG.download()
        for name, (o, shp) in oracle.resnet_layout(A).items():
            n = int(np.prod(shp))
            ref = grads_o[o:o + n]
            assert np.abs(g[o:o + n] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-7), (name, np.abs(g[o:o + n] - ref).max())
    finally:
        ctx.close()


@pytest.mark.parametrize("hid,channels", [([100], None), ([576], None), ([256, 256], None), (None, [16, 32, 64]), (None, [16, 32])])
def test_resnet_unsupported_widths_are_refused(hid, channels):
    with pytest.raises(RuntimeError, match="channels|hiddens"):
        L.Context(_cfg(hid, channels))


def test_cli_hiddens_trains_and_saves(tmp_path):
    """`--network impala_resnet --hiddens 128` end to end through train(): the host-env and device-env twins agree bit for bit, the parameter
    vector has the 128-wide layout, and the saved model reloads to the same vector."""
    import os
    from cleanba_amd.args import parse_args
    from cleanba_amd.checkpoint import load_cleanrl_model, save_cleanrl_model
    from cleanba_amd.trainer import train
    os.chdir(str(tmp_path))
    out = {}
    for backend in ("device", "host"):
        args = parse_args(["--local-num-envs", "8", "--num-actor-threads", "1", "--num-steps", "8", "--env-backend", backend, "--network",
                           "impala_resnet", "--hiddens", "128", "--total-timesteps", str(2 * 8 * 8), "--log-frequency", "1000"], "ppo")
        out[backend] = train(args, "ppo")
    p = out["device"]["params"]
    assert out["host"]["updates"] == out["device"]["updates"] == 2
    assert p.size == M.resnet_layout(A, 128)[1] and np.isfinite(p).all() and np.array_equal(p, out["host"]["params"])
    save_cleanrl_model(str(tmp_path / "m.cleanrl_model"), args, p, A, network="impala_resnet")
    _, q = load_cleanrl_model(str(tmp_path / "m.cleanrl_model"), A, network="impala_resnet")
    assert (bits(p) == bits(q)).all()


def test_cli_hiddens_save_model_then_evaluate(tmp_path, hidden_oracle):
    """`--hiddens 128 --save-model` (ppo:94, 753-785): the evaluation loop after training sizes its context from the SAVED vector's width
    (it used the default 256 and indexed the 128-wide vector with the 256-wide layout); replayed by the oracle at that width."""
    import os
    import cleanba_amd.prng as prng
    from cleanba_amd.args import parse_args
    from cleanba_amd.checkpoint import load_cleanrl_model
    from cleanba_amd.envs import make_env
    from cleanba_amd.evals import evaluate
    from cleanba_amd.trainer import train
    oracle = hidden_oracle
    os.chdir(str(tmp_path))
    args = parse_args(["--local-num-envs", "8", "--num-actor-threads", "1", "--num-steps", "8", "--env-backend", "device", "--network",
                       "impala_resnet", "--hiddens", "128", "--total-timesteps", "128", "--save-model", "--eval-episodes", "2",
                       "--eval-max-episode-steps", "24"], "ppo")
    res = train(args, "ppo")
    assert os.path.exists(res["model_path"]) and len(res["eval_returns"]) == 2
    _, params = load_cleanrl_model(res["model_path"], A, "impala_resnet")
    assert params.size == M.resnet_layout(A, 128)[1] and np.array_equal(params, res["params"])
    oracle.set_resnet_hidden(128)
    envs = make_env(args.env_id, 1, 1, backend="host")()
    key = prng.split(prng.prng_key(1), 4)[0]
    want = []
    for ep in range(2):
        obs, ret = envs.reset(), 0.0
        for _ in range(24):
            logits, _ = oracle.resnet_forward(params, A, obs, ksplit=11)
            a, _, key = oracle.sample_actions(logits, key)
            obs, _, _, info = envs.step(a)
            ret += float(info["reward"][0])
            if int(info["terminated"].sum()) + int(info["TimeLimit.truncated"].sum()) >= 1:
                break
        want.append(ret)
    assert res["eval_returns"] == want
    # a vector whose length is no layout at all is refused before anything is launched
    with open(res["model_path"], "rb") as f:
        blob = f.read()
    with pytest.raises(Exception):
        evaluate(res["model_path"], lambda e, s, n: make_env(e, s, n, backend="host"), args.env_id, 1, network="nature")
    assert blob
