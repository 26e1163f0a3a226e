"""A SECOND restatement of the scalar parts of the hot path — returns, loss heads, optimizers — in float64 numpy, written from the published
definitions of the pinned libraries (rlax 0.1.5 `vtrace_td_error_and_advantage`, optax 0.1.4 `clip_by_global_norm` / `scale_by_adam` /
`scale_by_rms` / `MultiSteps`, the GAE and PPO-clip papers) and from the reference's call sites, NOT from oracle/cbm_oracle.c: no shared
code, no shared intermediate expressions, gradients by central finite differences instead of hand-derived formulas.  Test infrastructure:
tests/test_numpy_ref.py holds the C oracle to this file, so two independently written implementations have to agree before either is
trusted as the checker of the HIP path."""
import numpy as np


def log_softmax(z):
    z = np.asarray(z, np.float64)
    s = z - z.max(-1, keepdims=True)
    return s - np.log(np.exp(s).sum(-1, keepdims=True))


# ------------------------------------------------------------------ returns
def gae(rewards, values, dones, next_value, next_done, gamma=0.99, lam=0.95):
    """Schulman et al. 2016 with the reference's done convention (ppo:532-560): dones[t] belongs to obs[t]; the step t -> t+1 is cut when
    dones[t+1] (next_done after the last step)."""
    r, v, d = (np.asarray(x, np.float64) for x in (rewards, values, dones))
    T = r.shape[0]
    v_next = np.concatenate([v[1:], np.asarray(next_value, np.float64)[None]], 0)
    cont = 1.0 - np.concatenate([d[1:], np.asarray(next_done, np.float64)[None]], 0)
    delta = r + gamma * v_next * cont - v
    adv = np.zeros_like(r)
    for t in range(T):          # forward definition: A_t = sum_k (gamma*lam)^k * prod(cont) * delta_{t+k}
        w = np.ones_like(r[0])
        for k in range(t, T):
            adv[t] += w * delta[k]
            w = w * gamma * lam * cont[k]
    return adv, adv + v


def vtrace(v_tm1, v_t, r_t, discount_t, rho_tm1, lam=1.0, clip_rho=1.0, clip_pg_rho=1.0):
    """rlax.vtrace_td_error_and_advantage (Espeholt et al. 2018, eq. 1) by its definition as a sum, not as the backward recursion:
    vs_t - V_t = sum_{k>=t} gamma-products * c-products * delta_k;  q_t = r_t + gamma_t * (lam * vs_{t+1} + (1 - lam) * V_{t+1})."""
    v_tm1, v_t, r_t, g, rho = (np.asarray(x, np.float64) for x in (v_tm1, v_t, r_t, discount_t, rho_tm1))
    T = r_t.shape[0]
    c = lam * np.minimum(1.0, rho)
    crho = np.minimum(clip_rho, rho)
    delta = crho * (r_t + g * v_t - v_tm1)
    err = np.zeros_like(r_t)
    for t in range(T):
        w = np.ones_like(r_t[0])
        for k in range(t, T):
            err[t] += w * delta[k]
            w = w * g[k] * c[k]
    target = err + v_tm1
    q_boot = np.concatenate([lam * target[1:] + (1.0 - lam) * v_tm1[1:], v_t[-1:]], 0)
    q = r_t + g * q_boot
    pg_adv = np.minimum(clip_pg_rho, rho) * (q - v_tm1)
    return target - v_tm1, pg_adv, q


# ------------------------------------------------------------------ losses (values only; gradients by finite differences)
def ppo_loss(logits, value, actions, old_logprob, adv, target, clip_coef=0.1, ent_coef=0.01, vf_coef=0.5):
    """ppo:516-530.  Returns (loss, pg_loss, v_loss, entropy, approx_kl)."""
    lp = log_softmax(logits)
    n = np.arange(lp.shape[0])
    new = lp[n, actions]
    logratio = new - np.asarray(old_logprob, np.float64)
    ratio = np.exp(logratio)
    adv = np.asarray(adv, np.float64)
    pg = np.maximum(-adv * ratio, -adv * np.clip(ratio, 1 - clip_coef, 1 + clip_coef)).mean()
    v = 0.5 * ((np.asarray(value, np.float64) - np.asarray(target, np.float64)) ** 2).mean()
    ent = -(np.exp(lp) * lp).sum(-1).mean()
    kl = ((ratio - 1) - logratio).mean()
    return pg - ent_coef * ent + vf_coef * v, pg, v, ent, kl


def impala_loss(logits, value, mu_logits, actions, rewards, dones, firststeps, gamma=0.99, vf_coef=0.5, ent_coef=0.01):
    """impala:547-597: [T+1, B] inputs; V-trace targets and advantages are constants of the differentiation (stop_gradient); the sums run
    over the T transitions whose source step is not an episode's first step."""
    logits, value, mu = (np.asarray(x, np.float64) for x in (logits, value, mu_logits))
    # impala:577-590: EVERY per-step input drops the bootstrap row with [:-1] (row t holds what arrived WITH obs_t, impala:372-384)
    mask = 1.0 - np.asarray(firststeps, np.float64)[:-1]
    disc = ((1.0 - np.asarray(dones, np.float64)) * gamma)[:-1]
    a = np.asarray(actions)[:-1]
    r = np.asarray(rewards, np.float64)[:-1]
    T, B = a.shape
    tt, bb = np.meshgrid(np.arange(T), np.arange(B), indexing="ij")
    lp = log_softmax(logits[:-1])
    lpa = lp[tt, bb, a]
    lma = log_softmax(mu[:-1])[tt, bb, a]
    rho = np.exp(lpa - lma)
    err, pg_adv, _ = vtrace(value[:-1], value[1:], r, disc, rho)
    return lpa, lp, err, pg_adv, mask


def impala_loss_value(logits, value, consts, vf_coef=0.5, ent_coef=0.01):
    """The differentiable part given the stop-gradient constants (targets = err + V, pg_adv) computed at the expansion point."""
    tgt, pg_adv, mask, a = consts
    logits, value = np.asarray(logits, np.float64), np.asarray(value, np.float64)
    T, B = a.shape
    tt, bb = np.meshgrid(np.arange(T), np.arange(B), indexing="ij")
    lp = log_softmax(logits[:-1])
    pg = (-(lp[tt, bb, a]) * pg_adv * mask).sum()
    bl = (0.5 * (tgt - value[:-1]) ** 2 * mask).sum()
    ent = ((np.exp(lp) * lp).sum(-1) * mask).sum()               # rlax.entropy_loss = -entropy: minimising it maximises entropy
    return pg + vf_coef * bl + ent_coef * ent, pg, bl, ent


def fd_grad(f, x, eps=1e-6):
    """Central finite differences of a scalar function of an array (float64)."""
    x = np.asarray(x, np.float64)
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        xp, xm = x.copy(), x.copy()
        xp[i] += eps
        xm[i] -= eps
        g[i] = (f(xp) - f(xm)) / (2 * eps)
    return g


# ------------------------------------------------------------------ optimizers
def clip_by_global_norm(g, max_norm):
    """optax 0.1.4: g_norm = global_norm(updates); trigger = g_norm < max_norm; updates = where(trigger, g, (g / g_norm) * max_norm)."""
    g = np.asarray(g, np.float64)
    n = np.sqrt((g ** 2).sum())
    return g if n < max_norm else (g / n) * max_norm


def adam(p, g, m, v, count, lr, b1=0.9, b2=0.999, eps=1e-5, max_norm=0.5):
    """optax.chain(clip_by_global_norm, inject_hyperparams(adam)(lr, eps=1e-5)) (ppo:492-500): scale_by_adam with bias correction by the
    incremented count, eps outside the square root, then scale(-lr)."""
    g = clip_by_global_norm(g, max_norm)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    t = count + 1
    mh, vh = m / (1 - b1 ** t), v / (1 - b2 ** t)
    return p - lr * mh / (np.sqrt(vh) + eps), m, v


def rmsprop(p, g, nu, lr, decay=0.99, eps=0.01, max_norm=40.0):
    """optax.chain(clip_by_global_norm, rmsprop(lr, eps=0.01, decay=0.99)) (impala:531-535): scale_by_rms(initial_scale=0): nu = decay*nu +
    (1-decay)*g^2; update = g / (sqrt(nu) + eps); then scale(-lr)."""
    g = clip_by_global_norm(g, max_norm)
    nu = decay * nu + (1 - decay) * g * g
    return p - lr * g / (np.sqrt(nu) + eps), nu


def multisteps_mean(grads):
    """optax.MultiSteps(every_k): acc <- acc + (g - acc) / (mini_step + 1); the k-th call hands the mean to the inner optimizer."""
    acc = np.zeros_like(np.asarray(grads[0], np.float64))
    for i, g in enumerate(grads):
        acc = acc + (np.asarray(g, np.float64) - acc) / (i + 1)
    return acc
