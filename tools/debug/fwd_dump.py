"""usage: CBM_SO=... python tools/debug/fwd_dump.py out.npy [B]  — logits+value of a seeded B-frame forward, twice (second run must equal the first)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import cleanba_amd.lib as L
from helpers import make_frames, make_params
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
cfg = L.default_config(L.ALGO_PPO); cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 120, 1, 128
ctx = L.Context(cfg)
P = make_params(18, 45); obs = make_frames(B, 44)
dP, dO = L.DevBuf(ctx, P), L.DevBuf(ctx, obs)
outs = []
for rep in range(3):
    dL = L.DevBuf(ctx, nbytes=B * 18 * 4, dtype=np.float32, shape=(B, 18)); dV = L.DevBuf(ctx, nbytes=B * 4, dtype=np.float32, shape=(B,))
    L._chk(ctx.lib.cbm_forward(ctx.h, L._p(dP.ptr), L._p(dO.ptr), None, B, 1, L._p(dL.ptr), L._p(dV.ptr)))
    outs.append(np.concatenate([dL.download().reshape(-1), dV.download()]))
print("repeatable:", all((o.view(np.uint32) == outs[0].view(np.uint32)).all() for o in outs))
np.save(sys.argv[1], outs[0])
