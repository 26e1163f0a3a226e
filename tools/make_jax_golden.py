#!/usr/bin/env python
"""Oracle pinning kit: regenerate golden vectors from the REAL reference stack.

    python tools/make_jax_golden.py            # needs jax 0.4.8, flax 0.6.8, optax 0.1.4, rlax 0.1.5 (requirements of vwxyzjn/cleanba)

Writes tests/golden/jax_vectors.npz: inputs (seeded numpy, identical to tests/golden/make_golden.py where they overlap) and the outputs
of jax / flax / optax / rlax on them.  tests/test_jax_golden.py then holds oracle/cbm_oracle.c to those outputs; while the file is absent
that test is skipped and the oracle stays "parity unpinned" (DESIGN.md section 3).

This image has no JAX and no network, so the script has never been run here; it is the kit to run the moment such an environment exists
(a build container — never the GPU box: nothing of jax / the reference travels with the repo, only the .npz data fixture does).  It covers
exactly SURVEY.md section 8c's "restated from knowledge of the pinned sources, unverified" ledger:

  prng      PRNGKey / split / uniform / permutation (the `_shuffle` round count and stable sort), ppo:468-470,599-606
  sample    Gumbel arg-max sampling + log-softmax of get_action_and_value, ppo:256-261
  div255    XLA keeps x / 255.0 a true division (naturecnn:159)
  forward   the flax Nature-CNN (naturecnn:143-178) on fixed parameters: logits / values
  gae       compute_gae's reverse lax.scan, ppo:532-560
  ppo       ppo_loss values and jax.grad w.r.t. logits / values, ppo:516-530
  vtrace    rlax.vtrace_td_error_and_advantage (lambda = 1, clips 1), impala:559-567
  impala    impala_loss values and jax.grad w.r.t. logits / values, impala:569-597
  adam      optax.chain(clip_by_global_norm, inject_hyperparams(adam)(linear_schedule, eps=1e-5)) — schedule count, clip '<', eps placement
  multi     optax.MultiSteps(every_k_schedule=2) running mean, ppo:492-500
  rmsprop   optax.chain(clip_by_global_norm(40), rmsprop(lr, eps=0.01, decay=0.99)), impala:531-535
  maxpool   flax.linen.max_pool(3x3, stride 2, SAME) pad split on 84 / 42 / 21 inputs, ppo:168-172
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

A = 18


def main():
    import flax
    import flax.linen as nn
    import jax
    import jax.numpy as jnp
    import optax
    import rlax
    from helpers import make_frames, make_params

    jax.config.update("jax_platform_name", "cpu")
    want = {"jax": "0.4.8", "flax": "0.6.8", "optax": "0.1.4", "rlax": "0.1.5"}
    have = {"jax": jax.__version__, "flax": flax.__version__, "optax": optax.__version__, "rlax": rlax.__version__}
    for k, v in want.items():
        if have[k] != v:
            print(f"WARNING: {k} {have[k]} is not the reference's pin {v}: the vectors pin THIS version's semantics", file=sys.stderr)
    rng = np.random.default_rng(2024)
    out = {"versions": np.array([f"{k}={v}" for k, v in have.items()])}

    # ---- prng
    key = jax.random.PRNGKey(1)
    out["seed1_key"] = np.asarray(key)
    out["seed1_split4"] = np.asarray(jax.random.split(key, 4))
    out["seed1_uniform16"] = np.asarray(jax.random.uniform(jax.random.split(key, 4)[0], (16,)))
    for n in (5, 257, 1000, 15360):
        out[f"perm_key7_n{n}"] = np.asarray(jax.random.permutation(jax.random.PRNGKey(7), n))
    out["bits_key7_n9"] = np.asarray(jax.random.bits(jax.random.PRNGKey(7), (9,), dtype=jnp.uint32))

    # ---- div255
    out["div255"] = np.asarray(jax.jit(lambda x: x / 255.0)(jnp.arange(256, dtype=jnp.float32)))

    # ---- forward: Nature-CNN of the legacy script (naturecnn:143-178) with OUR flat parameter layout unpacked into flax's tree
    class Network(nn.Module):
        @nn.compact
        def __call__(self, x):
            x = jnp.transpose(x, (0, 2, 3, 1))
            x = x / 255.0
            x = nn.Conv(32, kernel_size=(8, 8), strides=(4, 4), padding="VALID")(x)
            x = nn.relu(x)
            x = nn.Conv(64, kernel_size=(4, 4), strides=(2, 2), padding="VALID")(x)
            x = nn.relu(x)
            x = nn.Conv(64, kernel_size=(3, 3), strides=(1, 1), padding="VALID")(x)
            x = nn.relu(x)
            x = x.reshape((x.shape[0], -1))
            x = nn.Dense(512)(x)
            return nn.relu(x)

    P = make_params(A, 5)
    shapes = [("Conv_0", "kernel", (8, 8, 4, 32)), ("Conv_0", "bias", (32,)), ("Conv_1", "kernel", (4, 4, 32, 64)), ("Conv_1", "bias", (64,)),
              ("Conv_2", "kernel", (3, 3, 64, 64)), ("Conv_2", "bias", (64,)), ("Dense_0", "kernel", (3136, 512)), ("Dense_0", "bias", (512,))]
    tree, o = {}, 0
    for mod, leaf, shp in shapes:
        n = int(np.prod(shp))
        tree.setdefault(mod, {})[leaf] = jnp.asarray(P[o:o + n].reshape(shp))
        o += n
    aw, ab = P[o:o + 512 * A].reshape(512, A), P[o + 512 * A:o + 512 * A + A]
    o += 512 * A + A
    cw, cb = P[o:o + 512].reshape(512, 1), P[o + 512:o + 513]
    obs = make_frames(4, 6)
    hidden = Network().apply({"params": tree}, jnp.asarray(obs))
    out["fwd_obs"] = obs
    out["fwd_params_seed"] = np.int32(5)
    out["fwd_hidden"] = np.asarray(hidden)
    out["fwd_logits"] = np.asarray(hidden @ aw + ab)
    out["fwd_value"] = np.asarray((hidden @ cw + cb).squeeze(-1))

    # ---- sample (ppo:256-261)
    logits = jnp.asarray(out["fwd_logits"])
    skey = jax.random.PRNGKey(99)
    skey, subkey = jax.random.split(skey)
    u = jax.random.uniform(subkey, shape=logits.shape)
    action = jnp.argmax(logits - jnp.log(-jnp.log(u)), axis=1)
    logprob = jax.nn.log_softmax(logits)[jnp.arange(action.shape[0]), action]
    out.update(sample_actions=np.asarray(action), sample_logprob=np.asarray(logprob), sample_key_out=np.asarray(skey), sample_u=np.asarray(u))

    # ---- gae (ppo:532-560)
    T, B = 16, 8
    r = (rng.random((T, B)) < 0.1).astype(np.float32)
    v = rng.normal(size=(T, B)).astype(np.float32)
    d = (rng.random((T, B)) < 0.05).astype(np.float32)
    nv = rng.normal(size=B).astype(np.float32)
    nd = (rng.random(B) < 0.1).astype(np.float32)

    def compute_gae_once(carry, inp, gamma=0.99, gae_lambda=0.95):
        advantages = carry
        nextdone, nextvalues, curvalues, reward = inp
        nextnonterminal = 1.0 - nextdone
        delta = reward + gamma * nextvalues * nextnonterminal - curvalues
        advantages = delta + gamma * gae_lambda * nextnonterminal * advantages
        return advantages, advantages

    dones_ = jnp.concatenate([jnp.asarray(d), jnp.asarray(nd)[None, :]], axis=0)
    values_ = jnp.concatenate([jnp.asarray(v), jnp.asarray(nv)[None, :]], axis=0)
    _, adv = jax.lax.scan(compute_gae_once, jnp.zeros((B,)), (dones_[1:], values_[1:], values_[:-1], jnp.asarray(r)), reverse=True)
    out.update(gae_r=r, gae_v=v, gae_d=d.astype(np.uint8), gae_nv=nv, gae_nd=nd.astype(np.uint8), gae_adv=np.asarray(adv), gae_tgt=np.asarray(adv + values_[:-1]))
    a4 = np.asarray(adv).reshape(T, 4, -1)     # ppo:592-595: per env-column quarter normalisation
    out["gae_advnorm"] = ((a4 - a4.mean(axis=(0, 2), keepdims=True)) / (a4.std(axis=(0, 2), keepdims=True) + 1e-8)).reshape(T, B)
    jadv = jnp.asarray(adv).reshape(T, 4, -1)
    out["gae_advnorm_jnp"] = np.asarray(((jadv - jadv.mean((0, 2), keepdims=True)) / (jadv.std((0, 2), keepdims=True) + 1e-8)).reshape(T, B))

    # ---- ppo loss (ppo:516-530)
    N = 12
    lg = rng.normal(size=(N, A)).astype(np.float32)
    val = rng.normal(size=N).astype(np.float32)
    act = rng.integers(0, A, N).astype(np.int32)
    olp = (-np.log(A) + 0.3 * rng.normal(size=N)).astype(np.float32)
    ad = rng.normal(size=N).astype(np.float32)
    tg = rng.normal(size=N).astype(np.float32)

    def ppo_loss(lg_, val_, clip_coef=0.1, ent_coef=0.01, vf_coef=0.5):
        newlogprob = jax.nn.log_softmax(lg_)[jnp.arange(N), act]      # get_logprob_entropy_value, ppo:516-530
        z = lg_ - jax.scipy.special.logsumexp(lg_, axis=-1, keepdims=True)
        z = z.clip(min=jnp.finfo(z.dtype).min)
        entropy = -(z * jax.nn.softmax(z)).sum(-1)
        logratio = newlogprob - olp
        ratio = jnp.exp(logratio)
        approx_kl = ((ratio - 1) - logratio).mean()
        pg_loss = jnp.maximum(-ad * ratio, -ad * jnp.clip(ratio, 1 - clip_coef, 1 + clip_coef)).mean()
        v_loss = 0.5 * ((val_ - tg) ** 2).mean()
        entropy_loss = entropy.mean()
        return pg_loss - ent_coef * entropy_loss + vf_coef * v_loss, (pg_loss, v_loss, entropy_loss, approx_kl)

    (loss, aux), (dl, dv) = jax.value_and_grad(ppo_loss, argnums=(0, 1), has_aux=True)(jnp.asarray(lg), jnp.asarray(val))
    out.update(ppo_logits=lg, ppo_value=val, ppo_actions=act, ppo_oldlp=olp, ppo_adv=ad, ppo_tgt=tg,
               ppo_stats=np.array([loss, *aux], np.float32), ppo_dlogits=np.asarray(dl), ppo_dvalue=np.asarray(dv))

    # ---- vtrace + impala loss (impala:547-597)
    T1, Bm = 5, 3
    lg3 = rng.normal(size=(T1, Bm, A)).astype(np.float32)
    mu3 = rng.normal(size=(T1, Bm, A)).astype(np.float32)
    v3 = rng.normal(size=(T1, Bm)).astype(np.float32)
    a3 = rng.integers(0, A, (T1, Bm)).astype(np.int32)
    r3 = (rng.random((T1, Bm)) < 0.3).astype(np.float32)
    d3 = (rng.random((T1, Bm)) < 0.2).astype(np.float32)
    f3 = (rng.random((T1, Bm)) < 0.2).astype(np.float32)
    vt = jax.vmap(rlax.vtrace_td_error_and_advantage, in_axes=1, out_axes=1)

    def policy_gradient_loss(logits_, *args):
        mean_per_batch = jax.vmap(rlax.policy_gradient_loss, in_axes=1)(logits_, *args)
        return jnp.sum(mean_per_batch * logits_.shape[0])

    def entropy_loss_fn(logits_, *args):
        mean_per_batch = jax.vmap(rlax.entropy_loss, in_axes=1)(logits_, *args)
        return jnp.sum(mean_per_batch * logits_.shape[0])

    def impala_loss(policy_logits, newvalue, gamma=0.99, vf_coef=0.5, ent_coef=0.01):
        discounts = (1.0 - d3) * gamma
        mask = 1.0 - f3
        v_t, v_tm1 = newvalue[1:], newvalue[:-1]
        pl, ml, a_, mask_, rew_, disc_ = policy_logits[:-1], mu3[:-1], a3[:-1], mask[:-1], r3[:-1], discounts[:-1]
        rhos = rlax.categorical_importance_sampling_ratios(pl, ml, a_)
        ret = vt(v_tm1, v_t, rew_, disc_, rhos)
        pg_loss = policy_gradient_loss(pl, a_, ret.pg_advantage, mask_)
        baseline_loss = 0.5 * jnp.sum(jnp.square(ret.errors) * mask_)
        ent_loss = entropy_loss_fn(pl, mask_)
        return pg_loss + vf_coef * baseline_loss + ent_coef * ent_loss, (pg_loss, baseline_loss, ent_loss)

    (loss, aux), (dl, dv) = jax.value_and_grad(impala_loss, argnums=(0, 1), has_aux=True)(jnp.asarray(lg3), jnp.asarray(v3))
    out.update(imp_logits=lg3, imp_mu=mu3, imp_value=v3, imp_actions=a3, imp_rewards=r3, imp_dones=d3.astype(np.uint8), imp_first=f3.astype(np.uint8),
               imp_stats=np.array([loss, *aux], np.float32), imp_dlogits=np.asarray(dl), imp_dvalue=np.asarray(dv))
    Tn = 20
    V = rng.normal(size=(Tn + 1, 6)).astype(np.float32)
    rr = (rng.random((Tn, 6)) < 0.3).astype(np.float32)
    dd = (0.99 * (rng.random((Tn, 6)) > 0.1)).astype(np.float32)
    rh = np.exp(rng.normal(0, 0.5, size=(Tn, 6))).astype(np.float32)
    ret = vt(jnp.asarray(V[:-1]), jnp.asarray(V[1:]), jnp.asarray(rr), jnp.asarray(dd), jnp.asarray(rh))
    out.update(vt_V=V, vt_r=rr, vt_disc=dd, vt_rho=rh, vt_errors=np.asarray(ret.errors), vt_pg=np.asarray(ret.pg_advantage), vt_q=np.asarray(ret.q_estimate))

    # ---- optimizers
    n = 4096
    p0 = rng.normal(size=n).astype(np.float32)
    grads = [(s * rng.normal(size=n)).astype(np.float32) for s in (1e-4, 3.0, 1e-4, 3.0, 0.02, 0.02)]
    num_updates, spu = 10, 2                                   # linear_schedule ppo:475-479: frac = 1 - (count // steps_per_update) / num_updates

    def linear_schedule(count):
        frac = 1.0 - (count // spu) / num_updates
        return 2.5e-4 * frac

    tx = optax.chain(optax.clip_by_global_norm(0.5), optax.inject_hyperparams(optax.adam)(learning_rate=linear_schedule, eps=1e-5))
    st, p = tx.init(jnp.asarray(p0)), jnp.asarray(p0)
    traj, lrs = [], []
    for g in grads:
        upd, st = tx.update(jnp.asarray(g), st, p)
        lrs.append(float(st[1].hyperparams["learning_rate"]))
        p = optax.apply_updates(p, upd)
        traj.append(np.asarray(p))
    out.update(adam_p0=p0, adam_grads=np.stack(grads), adam_traj=np.stack(traj), adam_lrs=np.array(lrs, np.float32), adam_spu=np.int32(spu),
               adam_num_updates=np.int32(num_updates))
    txm = optax.MultiSteps(optax.chain(optax.clip_by_global_norm(0.5), optax.inject_hyperparams(optax.adam)(learning_rate=2.5e-4, eps=1e-5)), every_k_schedule=2)
    st, p = txm.init(jnp.asarray(p0)), jnp.asarray(p0)
    traj = []
    for g in grads[:4]:
        upd, st = txm.update(jnp.asarray(g), st, p)
        p = optax.apply_updates(p, upd)
        traj.append(np.asarray(p))
    out["multi_traj"] = np.stack(traj)
    txr = optax.chain(optax.clip_by_global_norm(40.0), optax.inject_hyperparams(optax.rmsprop)(learning_rate=6e-4, eps=0.01, decay=0.99))
    st, p = txr.init(jnp.asarray(p0)), jnp.asarray(p0)
    traj = []
    for g in grads:
        upd, st = txr.update(jnp.asarray(30 * g), st, p)
        p = optax.apply_updates(p, upd)
        traj.append(np.asarray(p))
    out["rms_traj"] = np.stack(traj)

    # ---- flax max_pool SAME (ppo:168-172)
    for hw in (84, 42, 21):
        x = rng.normal(size=(2, hw, hw, 3)).astype(np.float32)
        out[f"pool_in_{hw}"] = x
        out[f"pool_out_{hw}"] = np.asarray(nn.max_pool(jnp.asarray(x), window_shape=(3, 3), strides=(2, 2), padding="SAME"))

    path = os.path.join(ROOT, "tests", "golden", "jax_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(out), "arrays;", have)


if __name__ == "__main__":
    main()
