"""Timing build only (tools/variants.sh s16trace "-DCBM_S16_TRACE"; CBM_SO=cleanba_amd/abl_s16trace.so): the life of a block in each of the actor step's
four GEMM launches (igemm_s16_kernel at 120 frames, actor alone): first / middle / last block of the grid, wave 0, microseconds since the FIRST block's
entry — dispatch ramp, prologue, load latency, the K chunks, the store."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cleanba_amd.lib as L  # noqa: E402
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402

E, T, A = 120, 128, 18
cfg = L.default_config(L.ALGO_PPO)
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions, cfg.ring_depth = E, 1, T, A, 2
ctx = L.Context(cfg)
key = prng.prng_key(1)
key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_params("nature", A, nk, ak, ck))
ctx.actor_set_key(0, key)
ctx.actor_env_reset_device(0, 1)
for name, X, nchunk in (("conv1 (48000 x 32 x 256)", E * 400, 8), ("conv2 (9720 x 64 x 512)", E * 81, 8), ("conv3 (5880 x 64 x 576)", E * 49, 9), ("dense (120 x 512 x 224, 14 K-splits)", E, 7)):
    assert ctx.lib.cbm_debug_s16_trace(C.c_int(X), None) == 0
    acc, n = np.zeros((3, 24)), 0
    ctx.actor_begin_rollout(0, True)
    for k in range(T // 8):
        ctx.actor_rollout_device(0, 8)
        ctx.sync()
        buf = (C.c_uint64 * 72)()
        assert ctx.lib.cbm_debug_s16_trace(C.c_int(-1), buf) == 0
        t = np.array(buf, np.float64).reshape(3, 24)
        if k >= 2:
            acc += (t - t[0, 0]) / 2100.0
            n += 1
    ctx.actor_commit(0)
    ctx.params_publish_external(ctx.buffer("params")[0])
    acc /= n
    print(f"{name}: us since the first block's entry (first / middle / last block of the grid)")
    rows = [("entry", 0), ("prologue done", 1), ("first loads issued", 2), ("first tile in LDS", 3), ("first barrier", 4)] + [(f"chunk {c} done", 5 + c) for c in range(nchunk)] + [("stored, end", 22)]
    for nm, i in rows:
        print("  %-20s" % nm + "".join("%9.2f" % acc[b, i] for b in range(3)))
ctx.close()
