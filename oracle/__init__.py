"""ctypes front-end of the CPU oracle (oracle/cbm_oracle.c).

TEST INFRASTRUCTURE — see the header of cbm_oracle.c.  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  PARITY UNPINNED against JAX
(no jax/flax/optax/rlax here); pinned by known-answer vectors + torch-CPU autograd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcbm_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "cbm_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "cbm_math.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            build()   # no-op when the .so is newer than its sources
        except Exception:
            if not os.path.exists(_SO):
                raise
        _lib = C.CDLL(_SO)
        _lib.cbo_global_norm.restype = C.c_float
        _lib.cbo_logf.restype = C.c_float
        _lib.cbo_expf.restype = C.c_float
        _lib.cbo_u8_unit.restype = C.c_float
        _lib.cbo_nature_param_count.restype = C.c_int64
        for n in ("cbo_logf", "cbo_expf"):
            getattr(_lib, n).argtypes = [C.c_float]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _key(k):
    return np.ascontiguousarray(k, dtype=np.uint32)


FRAME = 4 * 84 * 84
ACT_PER_FRAME = 20 * 20 * 32 + 9 * 9 * 64 + 3136 + 512


def set_threads(n):
    lib().cbo_set_threads(int(n))


def set_conv1_exact(on):
    """conv1 of passes of more than 512 frames as the product's default computes it (cbm_config.conv1_fp32_chain bit 0 clear): exact uint8 x three-term-bf16
    products summed by the measured rule of v_mfma_f32_32x32x16_bf16 (cbm_oracle.c: cbo_mfma_bf16_group8) instead of the k-ascending fmaf chain.  Global, like
    set_threads; returns the previous setting."""
    prev = bool(lib().cbo_get_conv1_exact())
    lib().cbo_set_conv1_exact(int(bool(on)))
    return prev


def mfma_bf16_32x32x16(A, B, Cm):
    """D = A[32,16] x B[16,32] + C[32,32] as ONE v_mfma_f32_32x32x16_bf16 computes it; A / B are bf16 bit patterns (uint16)."""
    A = np.ascontiguousarray(A, np.uint16).reshape(32, 16)
    B = np.ascontiguousarray(B, np.uint16).reshape(16, 32)
    Cm = np.ascontiguousarray(Cm, np.float32).reshape(32, 32)
    D = np.zeros((32, 32), np.float32)
    lib().cbo_mfma_bf16_32x32x16(_p(A), _p(B), _p(Cm), _p(D))
    return D


# ---------------------------------------------------------------- PRNG
def threefry2x32(key, ctr):
    out = np.zeros(2, np.uint32)
    lib().cbo_threefry2x32(_p(_key(key)), _p(_key(ctr)), _p(out))
    return out


def prng_key(seed):
    out = np.zeros(2, np.uint32)
    lib().cbo_prng_key(C.c_uint64(int(seed)), _p(out))
    return out


def split(key, n=2):
    out = np.zeros((n, 2), np.uint32)
    lib().cbo_split(_p(_key(key)), int(n), _p(out))
    return out


def random_bits(key, n):
    out = np.zeros(n, np.uint32)
    lib().cbo_random_bits(_p(_key(key)), C.c_int64(n), _p(out))
    return out


def uniform(key, n):
    out = np.zeros(n, np.float32)
    lib().cbo_uniform(_p(_key(key)), C.c_int64(n), _p(out))
    return out


def permutation(key, n):
    out = np.zeros(n, np.int32)
    lib().cbo_permutation(_p(_key(key)), int(n), _p(out))
    return out


def logf(x):
    return float(lib().cbo_logf(C.c_float(x)))


def expf(x):
    return float(lib().cbo_expf(C.c_float(x)))


def logf_v(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().cbo_logf_v(_p(x), _p(y), C.c_int64(x.size))
    return y


def expf_v(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().cbo_expf_v(_p(x), _p(y), C.c_int64(x.size))
    return y


def bits_to_uniform_v(bits):
    b = np.ascontiguousarray(bits, np.uint32)
    y = np.empty(b.shape, np.float32)
    lib().cbo_bits_to_uniform_v(_p(b), _p(y), C.c_int64(b.size))
    return y


def u8_unit(x):
    return float(lib().cbo_u8_unit(C.c_uint32(int(x))))


# ---------------------------------------------------------------- network
def nature_param_count(A):
    return int(lib().cbo_nature_param_count(int(A)))


def nature_layout(A):
    """name -> (offset, shape) in the flat fp32 parameter blob (flax shapes)."""
    shapes = [("conv1.w", (8, 8, 4, 32)), ("conv1.b", (32,)), ("conv2.w", (4, 4, 32, 64)), ("conv2.b", (64,)),
              ("conv3.w", (3, 3, 64, 64)), ("conv3.b", (64,)), ("dense.w", (3136, 512)), ("dense.b", (512,)),
              ("actor.w", (512, A)), ("actor.b", (A,)), ("critic.w", (512, 1)), ("critic.b", (1,))]
    out, o = {}, 0
    for n, s in shapes:
        out[n] = (o, s)
        o += int(np.prod(s))
    assert o == nature_param_count(A)
    return out


def nature_forward(params, A, obs, idx=None, ksplit=1, save_acts=False):
    obs = _u8(obs)
    B = len(idx) if idx is not None else obs.shape[0]
    idx_a = _i32(idx) if idx is not None else None
    acts = np.zeros(B * ACT_PER_FRAME, np.float32) if save_acts else None
    logits = np.zeros((B, A), np.float32)
    value = np.zeros(B, np.float32)
    lib().cbo_nature_forward(_p(_f32(params)), int(A), _p(obs), _p(idx_a), int(B), int(ksplit), _p(acts),
                             _p(logits), _p(value))
    return (logits, value, acts) if save_acts else (logits, value)


def nature_backward(params, A, obs, idx, acts, dlogits, dvalue):
    obs = _u8(obs)
    B = dlogits.shape[0]
    idx_a = _i32(idx) if idx is not None else None
    grads = np.zeros(nature_param_count(A), np.float32)
    lib().cbo_nature_backward(_p(_f32(params)), int(A), _p(obs), _p(idx_a), int(B), _p(_f32(acts)),
                              _p(_f32(dlogits)), _p(_f32(dvalue)), _p(grads))
    return grads


def split_acts(acts, B):
    """[act1 NHWC | act2 | act3 | hid] views of the saved-activation blob."""
    o = 0
    out = []
    for shp in ((B, 20, 20, 32), (B, 9, 9, 64), (B, 3136), (B, 512)):
        n = int(np.prod(shp))
        out.append(acts[o:o + n].reshape(shp))
        o += n
    return out


def sample_actions(logits, key):
    logits = _f32(logits)
    B, A = logits.shape
    key_out = np.zeros(2, np.uint32)
    actions = np.zeros(B, np.int32)
    logprobs = np.zeros(B, np.float32)
    lib().cbo_sample_actions(_p(logits), B, A, _p(_key(key)), _p(key_out), _p(actions), _p(logprobs))
    return actions, logprobs, key_out


# ---------------------------------------------------------------- returns
def gae(rewards, values, dones, next_value, next_done, gamma=0.99, gae_lambda=0.95):
    rewards = _f32(rewards)
    T, B = rewards.shape
    adv = np.zeros((T, B), np.float32)
    tgt = np.zeros((T, B), np.float32)
    lib().cbo_gae(_p(rewards), _p(_f32(values)), _p(_u8(dones)), _p(_f32(next_value)), _p(_u8(next_done)), T, B,
                  C.c_float(gamma), C.c_float(gae_lambda), _p(adv), _p(tgt))
    return adv, tgt


def advnorm(adv, groups=4):
    a = _f32(adv).copy()
    T, B = a.shape
    lib().cbo_advnorm(_p(a), T, B, int(groups))
    return a


def gae_async(env_ids, rewards, values, dones, num_envs, gamma=0.99, gae_lambda=0.95):
    """prepare_data's reward re-index + env-id-indexed compute_gae of the legacy async script (naturecnn:232-262, 467-531)."""
    env_ids = _i32(env_ids)
    R, B = env_ids.shape
    adv = np.zeros((R, B), np.float32)
    tgt = np.zeros((R, B), np.float32)
    lib().cbo_gae_async(_p(env_ids), _p(_f32(rewards)), _p(_f32(values)), _p(_u8(dones)), R, B, int(num_envs), C.c_float(gamma),
                        C.c_float(gae_lambda), _p(adv), _p(tgt))
    return adv, tgt


def async_next_index(env_ids, num_envs):
    env_ids = _i32(env_ids).reshape(-1)
    out = np.zeros(env_ids.size, np.int32)
    lib().cbo_async_next_index(_p(env_ids), env_ids.size, int(num_envs), _p(out))
    return out


def mb_advnorm(adv):
    a = _f32(adv).reshape(-1)
    out = np.zeros_like(a)
    lib().cbo_mb_advnorm(_p(a), a.size, _p(out))
    return out


def vtrace(v_tm1, v_t, r_t, disc_t, rho_tm1):
    v_tm1 = _f32(v_tm1)
    T, B = v_tm1.shape
    e = np.zeros((T, B), np.float32)
    pg = np.zeros((T, B), np.float32)
    q = np.zeros((T, B), np.float32)
    lib().cbo_vtrace(_p(v_tm1), _p(_f32(v_t)), _p(_f32(r_t)), _p(_f32(disc_t)), _p(_f32(rho_tm1)), T, B, _p(e),
                     _p(pg), _p(q))
    return e, pg, q


# ---------------------------------------------------------------- losses
def ppo_loss_head(logits, value, actions, old_logprob, adv, target, clip_coef=0.1, ent_coef=0.01, vf_coef=0.5):
    logits = _f32(logits)
    N, A = logits.shape
    stats = np.zeros(5, np.float32)
    dlog = np.zeros((N, A), np.float32)
    dval = np.zeros(N, np.float32)
    lib().cbo_ppo_loss_head(_p(logits), _p(_f32(value)), N, A, _p(_i32(actions)), _p(_f32(old_logprob)),
                            _p(_f32(adv)), _p(_f32(target)), C.c_float(clip_coef), C.c_float(ent_coef),
                            C.c_float(vf_coef), _p(stats), _p(dlog), _p(dval))
    return stats, dlog, dval


def ppo_loss_grad(params, A, obs, idx, actions, old_logprob, adv, target, clip_coef=0.1, ent_coef=0.01, vf_coef=0.5,
                  ksplit=1, want_grads=True):
    obs = _u8(obs)
    N = len(actions)
    idx_a = _i32(idx) if idx is not None else None
    stats = np.zeros(5, np.float32)
    grads = np.zeros(nature_param_count(A), np.float32) if want_grads else None
    logits = np.zeros((N, A), np.float32)
    value = np.zeros(N, np.float32)
    lib().cbo_ppo_loss_grad(_p(_f32(params)), int(A), _p(obs), _p(idx_a), N, _p(_i32(actions)),
                            _p(_f32(old_logprob)), _p(_f32(adv)), _p(_f32(target)), C.c_float(clip_coef),
                            C.c_float(ent_coef), C.c_float(vf_coef), int(ksplit), _p(stats), _p(grads), _p(logits),
                            _p(value))
    return stats, grads, logits, value


def impala_loss_head(logits, value, mu_logits, actions, rewards, dones, firststeps, gamma=0.99, vf_coef=0.5,
                     ent_coef=0.01):
    logits = _f32(logits)
    T1, Bm, A = logits.shape
    stats = np.zeros(4, np.float32)
    dlog = np.zeros((T1, Bm, A), np.float32)
    dval = np.zeros((T1, Bm), np.float32)
    lib().cbo_impala_loss_head(_p(logits), _p(_f32(value)), _p(_f32(mu_logits)), _p(_i32(actions)),
                               _p(_f32(rewards)), _p(_u8(dones)), _p(_u8(firststeps)), T1, Bm, A, C.c_float(gamma),
                               C.c_float(vf_coef), C.c_float(ent_coef), _p(stats), _p(dlog), _p(dval))
    return stats, dlog, dval


def impala_loss_grad(params, A, obs, idx, T1, Bm, mu_logits, actions, rewards, dones, firststeps, gamma=0.99,
                     vf_coef=0.5, ent_coef=0.01):
    stats = np.zeros(4, np.float32)
    grads = np.zeros(nature_param_count(A), np.float32)
    idx_a = _i32(idx) if idx is not None else None
    lib().cbo_impala_loss_grad(_p(_f32(params)), int(A), _p(_u8(obs)), _p(idx_a), int(T1), int(Bm),
                               _p(_f32(mu_logits)), _p(_i32(actions)), _p(_f32(rewards)), _p(_u8(dones)),
                               _p(_u8(firststeps)), C.c_float(gamma), C.c_float(vf_coef), C.c_float(ent_coef),
                               _p(stats), _p(grads))
    return stats, grads


# ---------------------------------------------------------------- optimizers
def global_norm(g):
    g = _f32(g)
    return float(lib().cbo_global_norm(_p(g), C.c_int64(g.size)))


def adam_step(p, g, m, v, max_norm, lr, b1=0.9, b2=0.999, eps=1e-5, bc1=None, bc2=None, count=None):
    """In place on float32 arrays p, m, v.  bc1/bc2 default to 1-b^count in float32."""
    if bc1 is None:
        bc1 = np.float32(1.0) - np.power(np.float32(b1), np.float32(count))
        bc2 = np.float32(1.0) - np.power(np.float32(b2), np.float32(count))
    lib().cbo_adam_step(_p(p), _p(_f32(g)), _p(m), _p(v), C.c_int64(p.size), C.c_float(max_norm), C.c_float(lr),
                        C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(bc1), C.c_float(bc2))


def rmsprop_step(p, g, nu, max_norm, lr, decay=0.99, eps=0.01):
    lib().cbo_rmsprop_step(_p(p), _p(_f32(g)), _p(nu), C.c_int64(p.size), C.c_float(max_norm), C.c_float(lr),
                           C.c_float(decay), C.c_float(eps))


# ---------------------------------------------------------------- IMPALA-ResNet torso (ppo:149-189)
def set_resnet_hidden(h=256):
    """Width of the torso's one hidden layer (Network.hiddens, ppo:94) for every later resnet_* call; 256 = the reference default."""
    if lib().cbo_resnet_set_hidden(int(h)) != 0:
        raise ValueError(f"oracle: hidden width {h} outside [1, 512]")


def resnet_hidden():
    return int(lib().cbo_resnet_get_hidden())


def resnet_param_count(A):
    lib().cbo_resnet_param_count.restype = C.c_int64
    return int(lib().cbo_resnet_param_count(int(A)))


def resnet_act_floats():
    lib().cbo_resnet_act_floats.restype = C.c_int64
    return int(lib().cbo_resnet_act_floats())


def resnet_layout(A):
    """name -> (offset, shape), flax tree order: ConvSequence_s/{Conv_0, ResidualBlock_{0,1}/Conv_{0,1}}, Dense_0, heads."""
    ci, co, hd = (4, 16, 32), (16, 32, 32), resnet_hidden()
    out, o = {}, 0
    for s in range(3):
        names = ["Conv_0", "ResidualBlock_0.Conv_0", "ResidualBlock_0.Conv_1", "ResidualBlock_1.Conv_0", "ResidualBlock_1.Conv_1"]
        for j, n in enumerate(names):
            shp = (3, 3, ci[s] if j == 0 else co[s], co[s])
            out[f"seq{s}.{n}.w"] = (o, shp); o += int(np.prod(shp))
            out[f"seq{s}.{n}.b"] = (o, (co[s],)); o += co[s]
    for n, shp in (("dense.w", (3872, hd)), ("dense.b", (hd,)), ("actor.w", (hd, A)), ("actor.b", (A,)), ("critic.w", (hd, 1)),
                   ("critic.b", (1,))):
        out[n] = (o, shp); o += int(np.prod(shp))
    assert o == resnet_param_count(A)
    return out


def resnet_forward(params, A, obs, idx=None, ksplit=1, save_acts=False):
    obs = _u8(obs)
    B = len(idx) if idx is not None else obs.shape[0]
    idx_a = _i32(idx) if idx is not None else None
    acts = np.zeros(B * resnet_act_floats(), np.float32) if save_acts else None
    logits = np.zeros((B, A), np.float32)
    value = np.zeros(B, np.float32)
    lib().cbo_resnet_forward(_p(_f32(params)), int(A), _p(obs), _p(idx_a), int(B), int(ksplit), _p(acts), _p(logits), _p(value))
    return (logits, value, acts) if save_acts else (logits, value)


def resnet_backward(params, A, obs, idx, acts, dlogits, dvalue):
    B = dlogits.shape[0]
    idx_a = _i32(idx) if idx is not None else None
    grads = np.zeros(resnet_param_count(A), np.float32)
    lib().cbo_resnet_backward(_p(_f32(params)), int(A), _p(_u8(obs)), _p(idx_a), int(B), _p(_f32(acts)), _p(_f32(dlogits)), _p(_f32(dvalue)),
                              _p(grads))
    return grads
