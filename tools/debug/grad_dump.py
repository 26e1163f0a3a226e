"""usage: CBM_SO=... python tools/debug/grad_dump.py out.npy — flat gradient of a seeded 3840-frame PPO minibatch, three times."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cleanba_amd.lib as L
from helpers import make_frames, make_params
MB, A = 3840, 18
cfg = L.default_config(L.ALGO_PPO); cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 120, 1, 128
ctx = L.Context(cfg)
rng = np.random.default_rng(43)
P = make_params(A, 45); obs = make_frames(MB, 44)
dP, dO = L.DevBuf(ctx, P), L.DevBuf(ctx, obs)
idx = rng.permutation(MB).astype(np.int32); actions = rng.integers(0, A, MB).astype(np.int32)
lp = np.full(MB, -2.8, np.float32); adv = rng.normal(size=MB).astype(np.float32); tgt = rng.normal(size=MB).astype(np.float32)
d = [L.DevBuf(ctx, x) for x in (idx, actions, lp, adv, tgt)]
outs = []
for rep in range(3):
    dS = L.DevBuf(ctx, nbytes=32, dtype=np.float32, shape=(8,)); dG = L.DevBuf(ctx, nbytes=P.size * 4, dtype=np.float32, shape=(P.size,))
    L._chk(ctx.lib.cbm_ppo_loss_grad(ctx.h, L._p(dP.ptr), L._p(dO.ptr), L._p(d[0].ptr), MB, L._p(d[1].ptr), L._p(d[2].ptr), L._p(d[3].ptr), L._p(d[4].ptr),
                                      L._p(dS.ptr), L._p(dG.ptr), None, None))
    outs.append(dG.download())
print("repeatable:", [(o.view(np.uint32) == outs[0].view(np.uint32)).all() for o in outs], float(np.abs(outs[0]).max()))
np.save(sys.argv[1], outs[0])
