"""Independent torch-CPU (float64 by default) statement of the same math, used to cross-check
the C oracle: F.conv2d on NCHW-permuted flax weights + autograd for every gradient.
Follows the reference directly (naturecnn:143-178, ppo:516-577, impala:547-597), not the oracle."""
import numpy as np
import torch
import torch.nn.functional as F


def unpack_nature(params, A, dtype=torch.float64, requires_grad=False):
    shapes = [("conv1.w", (8, 8, 4, 32)), ("conv1.b", (32,)), ("conv2.w", (4, 4, 32, 64)), ("conv2.b", (64,)),
              ("conv3.w", (3, 3, 64, 64)), ("conv3.b", (64,)), ("dense.w", (3136, 512)), ("dense.b", (512,)),
              ("actor.w", (512, A)), ("actor.b", (A,)), ("critic.w", (512, 1)), ("critic.b", (1,))]
    flat = torch.tensor(np.asarray(params), dtype=dtype, requires_grad=requires_grad)
    out, o = {}, 0
    for n, s in shapes:
        k = int(np.prod(s))
        out[n] = flat[o:o + k].reshape(s)
        o += k
    return flat, out


def nature_forward(P, obs_u8):
    x = torch.tensor(np.asarray(obs_u8), dtype=P["conv1.w"].dtype) / 255.0           # NCHW
    def conv(x, w, b, s):
        return F.relu(F.conv2d(x, w.permute(3, 2, 0, 1), b, stride=s))                # HWIO -> OIHW
    x = conv(x, P["conv1.w"], P["conv1.b"], 4)
    x = conv(x, P["conv2.w"], P["conv2.b"], 2)
    x = conv(x, P["conv3.w"], P["conv3.b"], 1)
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                                  # (h,w,c) flatten
    h = F.relu(x @ P["dense.w"] + P["dense.b"])
    logits = h @ P["actor.w"] + P["actor.b"]
    value = (h @ P["critic.w"] + P["critic.b"]).squeeze(-1)
    return logits, value


def ppo_loss(logits, value, actions, old_lp, adv, target, clip=0.1, ent_coef=0.01, vf_coef=0.5):
    actions = torch.as_tensor(np.asarray(actions), dtype=torch.long)
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=logits.dtype)
    old_lp, adv, target = t(old_lp), t(adv), t(target)
    logp = F.log_softmax(logits, -1)
    newlp = logp.gather(1, actions[:, None]).squeeze(1)
    ent = -(logp * logp.exp()).sum(-1)
    logratio = newlp - old_lp
    ratio = logratio.exp()
    kl = ((ratio - 1) - logratio).mean()
    pg = torch.maximum(-adv * ratio, -adv * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
    v = 0.5 * ((value - target) ** 2).mean()
    e = ent.mean()
    return pg - ent_coef * e + vf_coef * v, (pg, v, e, kl)


def impala_loss(logits, value, mu_logits, actions, rewards, dones, firststeps, gamma=0.99, vf_coef=0.5, ent_coef=0.01):
    """logits [T1,B,A], value [T1,B]; rlax 0.1.5 vtrace with lambda=1 and clips 1."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=logits.dtype)
    mu_logits, rewards, dones, firststeps = t(mu_logits), t(rewards), t(dones), t(firststeps)
    actions = torch.as_tensor(np.asarray(actions), dtype=torch.long)
    disc = ((1.0 - dones) * gamma)[:-1]
    mask = (1.0 - firststeps)[:-1]
    v_t, v_tm1 = value[1:], value[:-1]
    pl, ml, a, r = logits[:-1], mu_logits[:-1], actions[:-1], rewards[:-1]
    lp = F.log_softmax(pl, -1)
    lpa = lp.gather(2, a[..., None]).squeeze(-1)
    lma = F.log_softmax(ml, -1).gather(2, a[..., None]).squeeze(-1)
    rho = (lpa - lma).exp().detach()
    c = torch.clamp(rho, max=1.0)
    td = c * (r + disc * v_t - v_tm1)
    T = td.shape[0]
    errs = [None] * T
    e = torch.zeros_like(td[0])
    for i in reversed(range(T)):
        e = td[i] + disc[i] * c[i] * e
        errs[i] = e
    err = torch.stack(errs)
    target = (err + v_tm1).detach()
    errors = target - v_tm1
    tg = errors + v_tm1
    qb = torch.cat([tg[1:], v_t[-1:]], 0)
    q = r + disc * qb
    pgadv = (c * (q - v_tm1)).detach()
    pg = (-lpa * pgadv * mask).sum()
    bl = 0.5 * (errors ** 2 * mask).sum()
    ent = (-(-(lp.exp() * lp).sum(-1)) * mask).sum()
    return pg + vf_coef * bl + ent_coef * ent, (pg, bl, ent)


def unpack_resnet(params, A, dtype=torch.float64, requires_grad=False, hid=256):
    ci, co = (4, 16, 32), (16, 32, 32)
    flat = torch.tensor(np.asarray(params), dtype=dtype, requires_grad=requires_grad)
    P, o = {}, 0
    def take(name, shp):
        nonlocal o
        k = int(np.prod(shp)); P[name] = flat[o:o + k].reshape(shp); o += k
    for s in range(3):
        for j in range(5):
            take(f"s{s}c{j}w", (3, 3, ci[s] if j == 0 else co[s], co[s])); take(f"s{s}c{j}b", (co[s],))
    take("dw", (3872, hid)); take("db", (hid,)); take("aw", (hid, A)); take("ab", (A,)); take("cw", (hid, 1)); take("cb", (1,))
    assert o == flat.numel()
    return flat, P


def resnet_forward(P, obs_u8):
    """ppo:149-189 with flax SAME semantics: conv pad 1; max_pool(3,3) stride 2 SAME = -inf pad (lo,hi) = (0,1),(0,1),(1,1)."""
    x = torch.tensor(np.asarray(obs_u8), dtype=P["dw"].dtype) / 255.0
    conv = lambda x, w, b: F.conv2d(x, w.permute(3, 2, 0, 1), b, padding=1)
    pads = [(0, 1), (0, 1), (1, 1)]
    for s in range(3):
        x = conv(x, P[f"s{s}c0w"], P[f"s{s}c0b"])
        lo, hi = pads[s]
        x = F.max_pool2d(F.pad(x, (lo, hi, lo, hi), value=float("-inf")), 3, 2)
        for blk in range(2):
            inp = x
            x = conv(F.relu(x), P[f"s{s}c{1 + 2 * blk}w"], P[f"s{s}c{1 + 2 * blk}b"])
            x = conv(F.relu(x), P[f"s{s}c{2 + 2 * blk}w"], P[f"s{s}c{2 + 2 * blk}b"])
            x = x + inp
    x = F.relu(x).permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    h = F.relu(x @ P["dw"] + P["db"])
    return h @ P["aw"] + P["ab"], (h @ P["cw"] + P["cb"]).squeeze(-1)
