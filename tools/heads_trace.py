"""Timing build only (CBM_SO=cleanba_amd/abl_tailtrace.so): clock stamps of block 0 of ppo_heads_fused_kernel (learner minibatches only, no actor),
per wave, microseconds since the block's first stamp."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cleanba_amd.lib as L  # noqa: E402
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402

E, T = 120, 128
cfg = L.default_config(L.ALGO_PPO)
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, 1, T
ctx = L.Context(cfg)
key = prng.prng_key(1)
key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_nature_params(18, nk, ak, ck))
ctx.actor_set_key(0, key)
ctx.actor_env_reset_device(0, 1)
ctx.actor_begin_rollout(0, False)
ctx.actor_rollout_device(0, T)
ctx.actor_commit(0)
ctx.learner_wait()
k = ctx.learner_prepare(key)
k = ctx.learner_epoch_begin(k)
names = ["entry", "staging issued", "sync A", "fragments read", "chain done", "sync B", "loss done", "sync C", "end"]
acc, n = np.zeros((4, 9)), 0
for i in range(12):
    ctx.learner_minibatch_grad(0, i % 4)
    ctx.sync()
    buf = (C.c_uint64 * 64)()
    assert ctx.lib.cbm_debug_tail_trace(buf) == 0
    t = np.array(buf, np.float64).reshape(4, 16)[:, :9]
    if i >= 2:
        d = (t - t[:, :1].min()) / 2100.0
        d[t == 0] = np.nan
        acc += d
        n += 1
acc /= n
print("ppo_heads_fused_kernel, block 0: us since its first stamp (avg of %d launches; waves 0-3)" % n)
for kk, nm in enumerate(names):
    print("%-16s" % nm + "".join(("%9.2f" % acc[w, kk]) if np.isfinite(acc[w, kk]) else "%9s" % "-" for w in range(4)))
ctx.close()
