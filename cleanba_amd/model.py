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


# ------------------------------------------------------------------ IMPALA-ResNet torso (ppo:149-189), the reference default
_RN_CI, _RN_CO = (4, 16, 32), (16, 32, 32)
_RN_CONVS = ("Conv_0", "ResidualBlock_0/Conv_0", "ResidualBlock_0/Conv_1", "ResidualBlock_1/Conv_0", "ResidualBlock_1/Conv_1")


def resnet_layout(A, hidden=256):
    """name -> (offset, shape), flax tree order (ConvSequence_s/{Conv_0, ResidualBlock_{0,1}/Conv_{0,1}}, Dense_0, heads).  `hidden` = the one
    hidden layer's width (Network.hiddens, ppo:94)."""
    out, o = {}, 0
    for s in range(3):
        for j, n in enumerate(_RN_CONVS):
            shp = (3, 3, _RN_CI[s] if j == 0 else _RN_CO[s], _RN_CO[s])
            out[f"ConvSequence_{s}/{n}/kernel"] = (o, shp); o += int(np.prod(shp))
            out[f"ConvSequence_{s}/{n}/bias"] = (o, (_RN_CO[s],)); o += _RN_CO[s]
    for n, shp in (("Dense_0/kernel", (3872, hidden)), ("Dense_0/bias", (hidden,)), ("actor/kernel", (hidden, A)), ("actor/bias", (A,)),
                   ("critic/kernel", (hidden, 1)), ("critic/bias", (1,))):
        out[n] = (o, shp); o += int(np.prod(shp))
    return out, o


def _lecun_normal(rng, shape):
    """flax default kernel_init (nn.Conv, ppo:156): truncated normal (+-2 sigma) with variance 1/fan_in."""
    fan_in = int(np.prod(shape[:-1]))
    std = np.sqrt(1.0 / fan_in) / 0.87962566103423978
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (std * x).astype(np.float32)


def init_resnet_params(A, network_key, actor_key, critic_key, hidden=256):
    layout, total = resnet_layout(A, hidden)
    p = np.zeros(total, np.float32)

    def rng_for(key, i):
        return np.random.Generator(np.random.Philox(key=[int(key[0]) << 32 | int(key[1]), i]))
    i = 0
    for name, (o, shp) in layout.items():
        if name.endswith("/kernel") and name.startswith("ConvSequence"):
            p[o:o + int(np.prod(shp))] = _lecun_normal(rng_for(network_key, i), shp).ravel()
            i += 1
    o, shp = layout["Dense_0/kernel"]
    p[o:o + int(np.prod(shp))] = _orthogonal(rng_for(network_key, 100), shp, np.sqrt(2.0)).ravel()
    o, shp = layout["actor/kernel"]
    p[o:o + int(np.prod(shp))] = _orthogonal(rng_for(actor_key, 0), shp, 0.01).ravel()
    o, shp = layout["critic/kernel"]
    p[o:o + int(np.prod(shp))] = _orthogonal(rng_for(critic_key, 0), shp, 1.0).ravel()
    return p


def init_params(network, A, network_key, actor_key, critic_key, hidden=256):
    if network == "nature":
        return init_nature_params(A, network_key, actor_key, critic_key)
    return init_resnet_params(A, network_key, actor_key, critic_key, hidden)


def resnet_hidden_of(P, A):
    """Hidden width of a flat ResNet parameter vector of length P (the layout is linear in it)."""
    _, p0 = resnet_layout(A, 0)
    h, r = divmod(P - p0, 3872 + 1 + A + 1)
    if r or h <= 0:
        raise ValueError(f"{P} parameters is not an IMPALA-ResNet with {A} actions")
    return h


def resnet_params_to_flax_tree(p, A):
    layout, _ = resnet_layout(A, resnet_hidden_of(len(p), A))
    net = {"params": {}}
    for name, (o, shp) in layout.items():
        parts = name.split("/")
        if parts[0] in ("actor", "critic"):
            continue
        d = net["params"]
        for k in parts[:-1]:
            d = d.setdefault(k, {})
        d[parts[-1]] = p[o:o + int(np.prod(shp))].reshape(shp).copy()

    def head(n):
        ow, sw = layout[f"{n}/kernel"]; ob, sb = layout[f"{n}/bias"]
        return {"params": {"Dense_0": {"kernel": p[ow:ow + int(np.prod(sw))].reshape(sw).copy(), "bias": p[ob:ob + sb[0]].copy()}}}
    return [net, head("actor"), head("critic")]


def resnet_flax_tree_to_params(tree, A):
    net, actor, critic = tree
    layout, total = resnet_layout(A, int(np.asarray(net["params"]["Dense_0"]["bias"]).shape[0]))
    p = np.zeros(total, np.float32)
    for name, (o, shp) in layout.items():
        parts = name.split("/")
        if parts[0] == "actor":
            a = actor["params"]["Dense_0"][parts[1]]
        elif parts[0] == "critic":
            a = critic["params"]["Dense_0"][parts[1]]
        else:
            a = net["params"]
            for k in parts:
                a = a[k]
        p[o:o + int(np.prod(shp))] = np.asarray(a, np.float32).reshape(shp).ravel()
    return p
