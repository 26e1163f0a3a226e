"""Timing build only (tools/variants.sh c1trace "-DCBM_CONV1_TRACE"; CBM_SO=cleanba_amd/abl_c1trace.so): clock stamps of block 0's SECOND frame in
conv1_fwd_planes_kernel (learner minibatch of 3840 frames, isolated), per wave: where a frame's ~25 k cycles go.  FRAMES=n limits the grid (fewer blocks: no second block on the CU)."""
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
acc, n = np.zeros((4, 27)), 0
for i in range(10):
    ctx.learner_minibatch_grad(0, i % 4)
    ctx.sync()
    buf = (C.c_uint64 * 128)()
    assert ctx.lib.cbm_debug_conv1_trace(buf) == 0
    t = np.array(buf, np.float64).reshape(4, 32)[:, :27]
    if i >= 2:
        acc += (t - t[:, :1].min())
        n += 1
acc /= n
print("conv1_fwd_planes_kernel, block 0, second frame: shader-clock cycles per phase (avg of %d launches; waves 0-3)" % n)
names = ["wait barrier 1", "convert + store + next loads", "wait barrier 2", "8 x 12 MFMA rows"]
tot = np.zeros(4)
for c in range(4):
    for ph in range(4):
        d = acc[:, 5 * c + ph + 1] - acc[:, 5 * c + ph]
        print("plane %d %-30s" % (c, names[ph]) + "".join("%9.0f" % x for x in d))
    if c < 3:
        print("plane %d -> %d gap                      " % (c, c + 1) + "".join("%9.0f" % x for x in (acc[:, 5 * (c + 1)] - acc[:, 5 * c + 4])))
print("%-38s" % "epilogue: wait for loads in flight" + "".join("%9.0f" % x for x in (acc[:, 22] - acc[:, 20])))
for t in range(3):
    print("%-38s" % ("epilogue: tile %d (16 stores + mask)" % t) + "".join("%9.0f" % x for x in (acc[:, 24 + t] - acc[:, 23 + t])))
print("%-38s" % "epilogue: tail tile" + "".join("%9.0f" % x for x in (acc[:, 21] - acc[:, 26])))
print("%-38s" % "epilogue (stores + mask)" + "".join("%9.0f" % x for x in (acc[:, 21] - acc[:, 20])))
print("%-38s" % "frame total" + "".join("%9.0f" % x for x in (acc[:, 21] - acc[:, 0])))
print("(96 x 3 = 288 32x32x2 MFMAs + 8 tail MFMAs per plane and wave = 18.4 k matrix-pipe cycles if alone on the SIMD; two blocks share a CU)")
ctx.close()
