"""The elastic conv1 weight gradient (slots whose remaining frames migrate to blocks that have finished, conv1.hip) must give the static kernel's
bits whatever migrates: runs three minibatch gradients beside a rollout and writes the flat gradients to argv[1] (.npy); run once with
CBM_C1W_ELASTIC=0 and once with the default, then `python tools/elastic_check.py --compare a.npy b.npy`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print("elastic == static bit for bit over %d gradients of %d floats: %s (max |diff| %.3g)" % (a.shape[0], a.shape[1], same, np.abs(a - b).max()))
    sys.exit(0 if same else 1)
import cleanba_amd.lib as L  # noqa: E402
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402

E, T, A = 120, 128, 18
cfg = L.default_config(L.ALGO_PPO)
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions = E, 1, T, A
ctx = L.Context(cfg)
key = prng.prng_key(1)
key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_nature_params(A, nk, ak, ck))
ctx.actor_set_key(0, key)
ctx.actor_env_reset_device(0, 1)
out = []
for rep in range(2):
    ctx.actor_begin_rollout(0, True); ctx.actor_rollout_device(0, T); ctx.actor_commit(0)
    ctx.sync()
    ctx.actor_begin_rollout(0, True); ctx.actor_rollout_device(0, T); ctx.actor_commit(0)    # this one runs beside the minibatches below
    ctx.learner_wait()
    k = ctx.learner_prepare(key)
    k = ctx.learner_epoch_begin(k)
    for mb in range(3):
        ctx.learner_minibatch_grad(0, mb)
        ctx.sync() if mb == 2 else None
        if mb == 2:
            out.append(ctx.read("grads", np.float32).copy())
    ctx.learner_finish(16, want_stats=False)
    ctx.sync()
    ctx.learner_wait()
    ctx.learner_prepare(key); ctx.learner_epoch_begin(k)
    ctx.learner_minibatch_grad(0, 0)
    ctx.sync()
    out.append(ctx.read("grads", np.float32).copy())
    ctx.learner_finish(16, want_stats=False)
    ctx.sync()
np.save(sys.argv[1], np.stack(out))
print("wrote", sys.argv[1], np.stack(out).shape)
