"""usage: [CBM_SO=...] python tools/debug/mask_check.py — after a 3840-frame forward, the ReLU bit masks must equal (activation > 0)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cleanba_amd.lib as L
from helpers import make_frames, make_params
B = 3840
cfg = L.default_config(L.ALGO_PPO); cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 120, 1, 128
ctx = L.Context(cfg)
dP, dO = L.DevBuf(ctx, make_params(18, 45)), L.DevBuf(ctx, make_frames(B, 44))
for rep in range(2):
    L._chk(ctx.lib.cbm_forward(ctx.h, L._p(dP.ptr), L._p(dO.ptr), None, B, 1, None, None))
    for name, ch in (("1", 32), ("2", 64), ("3", 64)):
        act = ctx.read("lws_act" + name, np.float32).reshape(-1, ch)
        m = ctx.read("lws_mask" + name, np.uint32).reshape(-1, ch // 32)
        rows = act.shape[0] if name != "1" else B * 400
        act, m = act[:rows], m[:rows]
        want = np.zeros_like(m)
        for w in range(ch // 32):
            bits = (act[:, 32 * w:32 * w + 32] > 0).astype(np.uint64)
            want[:, w] = (bits << np.arange(32, dtype=np.uint64)).sum(1).astype(np.uint32)
        bad = np.nonzero((want != m).any(1))[0]
        print("rep", rep, "mask" + name, "rows", rows, "bad rows", len(bad), bad[:12])
