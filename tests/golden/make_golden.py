#!/usr/bin/env python
"""Generates tests/golden/*.npz.  The reference cannot be imported here (no jax/flax/optax/rlax in the image, no
network), so these vectors come from the CPU oracle (oracle/cbm_oracle.c, the line-by-line
restatement) and serve as REGRESSION pins for both the oracle and the HIP path; the externally anchored pins are
the Random123 / JAX-documented known answers in tests/test_oracle_prng.py.  Re-run: `python tests/golden/make_golden.py`."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from helpers import make_frames, make_params  # noqa: E402

A = 18
rng = np.random.default_rng(2024)
out = {}
# PRNG streams at --seed 1 (ppo:468-470)
key = oracle.prng_key(1)
out["seed1_split4"] = oracle.split(key, 4)
out["seed1_uniform16"] = oracle.uniform(oracle.split(key, 4)[0], 16)
out["perm_key7_n257"] = oracle.permutation(oracle.prng_key(7), 257)
# forward + sampling on 4 frames (both dense K-split plans)
P = make_params(A, 5)
obs = make_frames(4, 6)
out["fwd_obs"] = obs
for ks in (1, 14):
    lg, v = oracle.nature_forward(P, A, obs, ksplit=ks)
    out[f"fwd_logits_ks{ks}"], out[f"fwd_value_ks{ks}"] = lg, v
a, lp, k2 = oracle.sample_actions(out["fwd_logits_ks14"], oracle.prng_key(99))
out["sample_actions"], out["sample_logprob"], out["sample_key_out"] = a, lp, k2
# returns
T, B = 16, 8
r = (rng.random((T, B)) < 0.1).astype(np.float32)
v = rng.normal(size=(T, B)).astype(np.float32)
d = (rng.random((T, B)) < 0.05).astype(np.uint8)
nv = rng.normal(size=B).astype(np.float32)
nd = (rng.random(B) < 0.1).astype(np.uint8)
adv, tgt = oracle.gae(r, v, d, nv, nd)
out.update(gae_r=r, gae_v=v, gae_d=d, gae_nv=nv, gae_nd=nd, gae_adv=adv, gae_tgt=tgt, gae_advnorm=oracle.advnorm(adv, 4))
# PPO / IMPALA loss heads on fixed logits
N = 12
lg = rng.normal(size=(N, A)).astype(np.float32)
val = rng.normal(size=N).astype(np.float32)
act = rng.integers(0, A, N).astype(np.int32)
olp = (-np.log(A) + 0.3 * rng.normal(size=N)).astype(np.float32)
ad = rng.normal(size=N).astype(np.float32)
tg = rng.normal(size=N).astype(np.float32)
st, dl, dv = oracle.ppo_loss_head(lg, val, act, olp, ad, tg)
out.update(ppo_logits=lg, ppo_value=val, ppo_actions=act, ppo_oldlp=olp, ppo_adv=ad, ppo_tgt=tg, ppo_stats=st, ppo_dlogits=dl, ppo_dvalue=dv)
T1, Bm = 5, 3
lg3 = rng.normal(size=(T1, Bm, A)).astype(np.float32)
mu3 = rng.normal(size=(T1, Bm, A)).astype(np.float32)
v3 = rng.normal(size=(T1, Bm)).astype(np.float32)
a3 = rng.integers(0, A, (T1, Bm)).astype(np.int32)
r3 = (rng.random((T1, Bm)) < 0.3).astype(np.float32)
d3 = (rng.random((T1, Bm)) < 0.2).astype(np.uint8)
f3 = (rng.random((T1, Bm)) < 0.2).astype(np.uint8)
st, dl, dv = oracle.impala_loss_head(lg3, v3, mu3, a3, r3, d3, f3)
out.update(imp_logits=lg3, imp_mu=mu3, imp_value=v3, imp_actions=a3, imp_rewards=r3, imp_dones=d3, imp_first=f3, imp_stats=st, imp_dlogits=dl,
           imp_dvalue=dv)
np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
print("wrote", os.path.join(HERE, "oracle_vectors.npz"), {k: v.shape for k, v in out.items() if k.startswith(("fwd", "ppo_stats", "imp_stats"))})

# legacy --async-batch-size returns: env-id-indexed GAE + per-minibatch advantage normalisation, own file
from test_oracle_async import make_async_rollout  # noqa: E402
asy = {}
for tag, (R, B, NE, seed) in {"a": (24, 3, 7, 11), "b": (128, 20, 60, 12)}.items():
    env_ids, r, v, d = make_async_rollout(R, B, NE, seed)
    adv, tgt = oracle.gae_async(env_ids, r, v, d, NE)
    asy.update({f"{tag}_env_ids": env_ids, f"{tag}_r": r, f"{tag}_v": v, f"{tag}_d": d, f"{tag}_adv": adv, f"{tag}_tgt": tgt,
                f"{tag}_num_envs": np.int32(NE), f"{tag}_next_index": oracle.async_next_index(env_ids, NE),
                f"{tag}_mbnorm": oracle.mb_advnorm(adv.reshape(-1)[: R * B // 2])})
np.savez_compressed(os.path.join(HERE, "async_vectors.npz"), **asy)
print("wrote", os.path.join(HERE, "async_vectors.npz"))
