"""Phase timeline of impala_loss_kernel (block 0, thread 0, s_memtime stamps at the phase barriers).  Needs the timing build:
  tools/variants.sh iltrace "-DCBM_IL_TRACE"; CBM_SO=$PWD/cleanba_amd/abl_iltrace.so python tools/il_trace.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import cleanba_amd.lib as L, cleanba_amd.model as M, cleanba_amd.prng as prng
E, T, A = 120, 128, 18
cfg = L.default_config(L.ALGO_IMPALA)
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions = E, 1, T, A
ctx = L.Context(cfg)
key = prng.prng_key(1); key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_nature_params(A, nk, ak, ck)); ctx.actor_set_key(0, key); ctx.actor_env_reset_device(0, 1)
ctx.actor_begin_rollout(0, True); ctx.actor_rollout_device(0, T + 1); ctx.actor_commit(0); ctx.learner_wait()
names = ["entry", "operands in LDS", "row maxima", "exponentials", "sums, logs", "scan inputs, p log p", "scan | entropy", "loss terms", "gradients stored"]
acc, n = np.zeros(9), 0
for i in range(8):
    ctx.learner_minibatch_grad(0, i % 4); ctx.sync()
    buf = (C.c_uint64 * 16)(); assert ctx.lib.cbm_debug_il_trace(buf) == 0
    t = np.array(buf, np.float64)[:9]
    if i >= 2: acc += (t - t[0]) / 2100.0; n += 1
for k, nm in enumerate(names): print("%-14s %8.2f us" % (nm, acc[k] / n))
ctx.close()
