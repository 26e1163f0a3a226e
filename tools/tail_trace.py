"""Timing build only (tools/variants.sh <name> "-DCBM_TAIL_TRACE"; CBM_SO=cleanba_amd/abl_<name>.so): shader-clock stamps of block 0 of the last
actor_tail_rows_kernel launch of a rollout, per wave, as microseconds since the block's first stamp (2.1 GHz assumed) — where the tail's time goes."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cleanba_amd.lib as L  # noqa: E402
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402

E, T, A = 120, 128, 18
cfg = L.default_config(L.ALGO_PPO if os.environ.get("ALGO", "ppo") == "ppo" else L.ALGO_IMPALA)
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions, cfg.ring_depth = E, 1, T, A, 2
ctx = L.Context(cfg)
key = prng.prng_key(1)
key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_params("nature", A, nk, ak, ck))
ctx.actor_set_key(0, key)
ctx.actor_env_reset_device(0, 1)
names = ["entry", "loads issued", "candidates", "reduce done", "sync1", "early", "mfma", "sync2", "sampled", "sync3", "end", "prefetch issd", "vv issued", "B reads issd", "h reads issd"]
order = [0, 11, 12, 1, 2, 3, 4, 13, 14, 6, 5, 7, 8, 9, 10]
acc = np.zeros((4, 15))
n = 0
for rep in range(6):
    ctx.actor_begin_rollout(0, True)
    for k in range(T // 8):
        ctx.actor_rollout_device(0, 8)
        ctx.sync()
        buf = (C.c_uint64 * 64)()
        assert ctx.lib.cbm_debug_tail_trace(buf) == 0
        t = np.array(buf, np.float64).reshape(4, 16)[:, :15]
        if rep >= 1:
            d = (t - t[:, :1].min()) / 2100.0
            d[t == 0] = np.nan                 # a stamp this wave never takes (only waves 0 / 1 run the heads' chain)
            acc += d
            n += 1
    ctx.actor_commit(0)
    ctx.params_publish_external(ctx.buffer("params")[0])
acc /= n
print("us since the block's first stamp (avg of %d launches; rows = waves 0-3)" % n)
print("%-14s" % "" + "".join("%9s" % ("w%d" % w) for w in range(4)))
for k in order:
    print("%-14s" % names[k] + "".join(("%9.2f" % acc[w, k]) if np.isfinite(acc[w, k]) else "%9s" % "-" for w in range(4)))
ctx.close()
