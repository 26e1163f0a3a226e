"""Timing build only (tools/variants.sh blocktrace "-DCBM_BLOCK_TRACE"; CBM_SO=cleanba_amd/abl_blocktrace.so): start / end wall-clock stamps of EVERY
block of the last launch of conv1 forward, conv1 weight gradient (persistent, static frame ranges) and the dense forward (one wave of 480 tiles) —
alone and beside the rollout.  Question: is a learner kernel's stretch under the rollout UNEVEN across its blocks (then handing work out dynamically
would recover it) or uniform (then it would not)?"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cleanba_amd.lib as L  # noqa: E402
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402

E, T, A, EPOCHS, NMB = 120, 128, 18, 4, 4
cfg = L.default_config(L.ALGO_PPO)
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions = E, 1, T, A
ctx = L.Context(cfg)
key = prng.prng_key(1)
key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_nature_params(A, nk, ak, ck))
ctx.actor_set_key(0, key)
ctx.actor_env_reset_device(0, 1)
lkey = key.copy()
n_opt = EPOCHS * NMB
lrs = np.full(n_opt, 2.5e-4, np.float32)
bc = [M.adam_bias_corrections(i + 1) for i in range(n_opt)]
b1 = np.array([b[0] for b in bc], np.float32)
b2 = np.array([b[1] for b in bc], np.float32)


def rollout():
    ctx.actor_begin_rollout(0, True); ctx.actor_rollout_device(0, T); ctx.actor_commit(0)


def update():
    global lkey
    ctx.learner_wait()
    lkey, _ = ctx.learner_update(lkey, lrs, b1, b2, want_stats=False)


def read():
    a = (C.c_uint64 * (2 * 512 * 2))()
    assert ctx.lib.cbm_debug_block_trace_conv1(a) == 0
    c1 = np.array(a, np.float64).reshape(2, 512, 2)
    d = (C.c_uint64 * (512 * 2))()
    assert ctx.lib.cbm_debug_block_trace_dma(d) == 0
    return {"conv1_fwd": c1[0], "conv1_wgrad": c1[1], "dense_fwd": np.array(d, np.float64).reshape(512, 2)}


def report(tag, tr):
    for name, t in tr.items():
        t = t[t[:, 1] > 0]
        st, en = t[:, 0] / 100.0, t[:, 1] / 100.0   # us
        dur = en - st
        span = en.max() - st.min()
        print("%-10s %-12s blocks %3d  kernel span %7.1f us | block life: mean %7.1f  p5 %7.1f  p95 %7.1f  max %7.1f | last start +%6.1f us | "
              "ends: p5 %7.1f  p50 %7.1f  p95 %7.1f (from first start) | idle tail = 1 - mean(end)/span = %.3f"
              % (tag, name, len(t), span, dur.mean(), np.percentile(dur, 5), np.percentile(dur, 95), dur.max(), st.max() - st.min(),
                 np.percentile(en - st.min(), 5), np.percentile(en - st.min(), 50), np.percentile(en - st.min(), 95), 1.0 - (en - st.min()).mean() / span))


# alone: one update with nothing else on the GPU
rollout(); ctx.sync()
rollout()
update(); ctx.sync()
report("alone", read())
# beside the rollout: a rollout is enqueued (640 launches, ~6.5 ms alone), then the first minibatches of an update through the split API; the stamps
# read back are those of minibatch 2, launched ~4 ms into the rollout
for rep in range(3):
    rollout()                      # (rollout u+1: its parameters exist, its ring entry is free)
    ctx.learner_wait()
    k = ctx.learner_prepare(lkey)
    k = ctx.learner_epoch_begin(k)
    for mb in range(3):
        ctx.learner_minibatch_grad(0, mb)
    ctx.sync()
    report("beside", read())
    ctx.learner_finish(n_opt, want_stats=False)     # (no optimizer steps were taken: the stamps are all this loop is for) version + ring entry advance
    ctx.sync()
