"""Evaluation loop — counterpart of `cleanrl_utils/evals/ppo_envpool_jax_eval.py:13-82` (called after `--save-model`, ppo:773-782).

Same contract: load a `.cleanrl_model`, one environment, SAMPLED actions through the Gumbel-max rule with the key chain
`key = PRNGKey(seed); key, *_ = split(key, 4)` then `key, subkey = split(key)` per step (eval.py:30-31,52-55), an episode ends on
`terminated` or `TimeLimit.truncated`, the return is the sum of the UNCLIPPED `info["reward"]`.  Inference runs on the HIP library
(one-env actor slot); video capture is out of scope (no moviepy/cv2 here).
"""
import numpy as np

from . import lib as L
from . import model as M
from . import prng
from .checkpoint import load_cleanrl_model


def evaluate(model_path, make_env, env_id, eval_episodes, run_name=None, Model=None, capture_video=False, seed=1, network=None,
             max_episode_steps=None):
    envs = make_env(env_id, seed, 1)()
    num_actions = envs.single_action_space.n
    if network is None:
        import msgpack
        with open(model_path, "rb") as f:   # the saved vars(args) carry the torso kind
            saved_args = msgpack.unpackb(f.read(), ext_hook=lambda c, d: None, raw=False)["0"]
        network = saved_args.get("network", "impala_resnet")
    _, params = load_cleanrl_model(model_path, num_actions, network)
    cfg = L.default_config(L.ALGO_PPO)
    cfg.network = L.NET_NATURE if network == "nature" else L.NET_IMPALA_RESNET
    cfg.num_actions, cfg.local_num_envs, cfg.num_actor_slots = num_actions, 1, 1
    cfg.num_steps, cfg.num_minibatches, cfg.update_epochs = 8, 1, 1
    ksplit = cfg.actor_dense_ksplit = 14 if network == "nature" else 11   # the actor's numerics
    if network != "nature":   # --hiddens H (ppo:94): the width is a property of the saved parameter vector
        cfg.num_hiddens, cfg.hiddens[0] = 1, M.resnet_hidden_of(len(params), num_actions)
    ctx = L.Context(cfg)
    if len(params) != ctx.P:
        ctx.close()
        raise ValueError(f"{model_path}: {len(params)} parameters, the {network} layout with {num_actions} actions has {ctx.P}")
    d_params = L.DevBuf(ctx, np.ascontiguousarray(params, np.float32))
    d_obs = L.DevBuf(ctx, nbytes=L.FRAME, dtype=np.uint8)
    d_logits = L.DevBuf(ctx, nbytes=num_actions * 4, dtype=np.float32)
    d_value = L.DevBuf(ctx, nbytes=4, dtype=np.float32)
    d_action = L.DevBuf(ctx, nbytes=4, dtype=np.int32)
    d_logprob = L.DevBuf(ctx, nbytes=4, dtype=np.float32)
    key = prng.split(prng.prng_key(seed), 4)[0]
    limit = max_episode_steps or envs.spec.config.max_episode_steps
    episodic_returns = []
    for episode in range(eval_episodes):
        episodic_return = 0.0
        next_obs = envs.reset()
        for _ in range(limit):
            # get_action_and_value of eval.py:42-56: forward, key split, Gumbel arg-max
            d_obs.upload(next_obs)
            L._chk(ctx.lib.cbm_forward(ctx.h, L._p(d_params.ptr), L._p(d_obs.ptr), None, 1, ksplit, L._p(d_logits.ptr), L._p(d_value.ptr)))
            ks = prng.split(key, 2)
            key, subkey = ks[0], np.ascontiguousarray(ks[1], np.uint32)
            L._chk(ctx.lib.cbm_sample(ctx.h, L._p(d_logits.ptr), 1, L._p(subkey), L._p(d_action.ptr), L._p(d_logprob.ptr)))
            actions = d_action.download()
            next_obs, _, _, infos = envs.step(actions)
            episodic_return += float(infos["reward"][0])
            if int(np.sum(infos["terminated"])) + int(np.sum(infos["TimeLimit.truncated"])) >= 1:
                break
        print(f"eval_episode={len(episodic_returns)}, episodic_return={episodic_return}")
        episodic_returns.append(episodic_return)
    for b in (d_params, d_obs, d_logits, d_value, d_action, d_logprob):
        b.free()
    ctx.close()
    envs.close()
    return episodic_returns
