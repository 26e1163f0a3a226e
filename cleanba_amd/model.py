"""Host-side model plumbing: parameter layout (flax names), initialisation, schedules, checkpoints.

Reference: Network/Actor/Critic definitions naturecnn:143-178 + ppo:192-203, init ppo:481-501,
linear_schedule ppo:475-479 / impala:515-519, optimizer bias corrections (optax 0.1.4 scale_by_adam).
"""
import numpy as np

from . import lib as L


def nature_layout(A):
    """name -> (offset, shape): flat fp32 blob in flax shapes (conv HWIO, dense [in,out])."""
    shapes = [("conv1.w", (8, 8, 4, 32)), ("conv1.b", (32,)), ("conv2.w", (4, 4, 32, 64)), ("conv2.b", (64,)),
              ("conv3.w", (3, 3, 64, 64)), ("conv3.b", (64,)), ("dense.w", (3136, 512)), ("dense.b", (512,)),
              ("actor.w", (512, A)), ("actor.b", (A,)), ("critic.w", (512, 1)), ("critic.b", (1,))]
    out, o = {}, 0
    for n, s in shapes:
        out[n] = (o, s)
        o += int(np.prod(s))
    return out, o


def _orthogonal(rng, shape, scale):
    """flax.linen.initializers.orthogonal(scale): QR of a normal matrix with sign fix (column_axis=-1)."""
    n_cols = shape[-1]
    n_rows = int(np.prod(shape)) // n_cols
    a = rng.standard_normal((max(n_rows, n_cols), min(n_rows, n_cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if n_rows < n_cols:
        q = q.T
    return (scale * q.reshape(shape)).astype(np.float32)


def init_nature_params(A, network_key, actor_key, critic_key):
    """Nature-CNN: every kernel orthogonal(sqrt 2), heads orthogonal(0.01) / orthogonal(1), biases 0
    (naturecnn:147-176, ppo:195,203).  The normal draws come from numpy Philox streams keyed by the
    jax-compatible keys; flax's own normal/QR bit patterns are not reproducible without jax — the
    distribution is, and checkpoints (save/load below) carry exact values across frameworks."""
    layout, total = nature_layout(A)
    p = np.zeros(total, np.float32)

    def rng_for(key, i):
        return np.random.Generator(np.random.Philox(key=[int(key[0]) << 32 | int(key[1]), i]))

    for i, name in enumerate(("conv1.w", "conv2.w", "conv3.w", "dense.w")):
        o, s = layout[name]
        p[o:o + int(np.prod(s))] = _orthogonal(rng_for(network_key, i), s, np.sqrt(2.0)).ravel()
    o, s = layout["actor.w"]
    p[o:o + int(np.prod(s))] = _orthogonal(rng_for(actor_key, 0), s, 0.01).ravel()
    o, s = layout["critic.w"]
    p[o:o + int(np.prod(s))] = _orthogonal(rng_for(critic_key, 0), s, 1.0).ravel()
    return p


def linear_schedule(count, learning_rate, steps_per_update, num_updates, anneal=True):
    """ppo:475-479 / impala:515-519 in float32: lr * (1 - (count // steps_per_update) / num_updates)."""
    if not anneal:
        return np.float32(learning_rate)
    frac = np.float32(1.0) - np.float32(count // steps_per_update) / np.float32(num_updates)
    return np.float32(learning_rate) * frac


def adam_bias_corrections(count_inc, b1=0.9, b2=0.999):
    """optax.scale_by_adam: 1 - b^count with count already incremented (float32 pow)."""
    c = np.float32(count_inc)
    return (np.float32(1.0) - np.power(np.float32(b1), c), np.float32(1.0) - np.power(np.float32(b2), c))


# ---- .cleanrl_model-compatible parameter trees: flax names <-> flat blob
def params_to_flax_tree(p, A):
    layout, _ = nature_layout(A)

    def get(n):
        o, s = layout[n]
        return p[o:o + int(np.prod(s))].reshape(s).copy()
    network = {"params": {"Conv_0": {"kernel": get("conv1.w"), "bias": get("conv1.b")},
                          "Conv_1": {"kernel": get("conv2.w"), "bias": get("conv2.b")},
                          "Conv_2": {"kernel": get("conv3.w"), "bias": get("conv3.b")},
                          "Dense_0": {"kernel": get("dense.w"), "bias": get("dense.b")}}}
    actor = {"params": {"Dense_0": {"kernel": get("actor.w"), "bias": get("actor.b")}}}
    critic = {"params": {"Dense_0": {"kernel": get("critic.w"), "bias": get("critic.b")}}}
    return [network, actor, critic]


def flax_tree_to_params(tree, A):
    layout, total = nature_layout(A)
    network, actor, critic = tree
    p = np.zeros(total, np.float32)

    def put(n, a):
        o, s = layout[n]
        p[o:o + int(np.prod(s))] = np.asarray(a, np.float32).reshape(s).ravel()
    for i, n in enumerate(("conv1", "conv2", "conv3")):
        put(n + ".w", network["params"][f"Conv_{i}"]["kernel"])
        put(n + ".b", network["params"][f"Conv_{i}"]["bias"])
    put("dense.w", network["params"]["Dense_0"]["kernel"])
    put("dense.b", network["params"]["Dense_0"]["bias"])
    put("actor.w", actor["params"]["Dense_0"]["kernel"])
    put("actor.b", actor["params"]["Dense_0"]["bias"])
    put("critic.w", critic["params"]["Dense_0"]["kernel"])
    put("critic.b", critic["params"]["Dense_0"]["bias"])
    return p
