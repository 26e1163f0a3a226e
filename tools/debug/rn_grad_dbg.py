import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cleanba_amd.lib as L
import oracle
from helpers import make_frames
from test_oracle_resnet import make_resnet_params
A = 18
cfg = L.default_config(L.ALGO_PPO)
cfg.network = L.NET_IMPALA_RESNET
cfg.actor_dense_ksplit = 11
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 8, 1, 8
rctx = L.Context(cfg)
rng = np.random.default_rng(6)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = make_resnet_params(oracle, 7)
obs = make_frames(24, 8)
idx = rng.permutation(24)[:N].astype(np.int32)
actions = rng.integers(0, A, N).astype(np.int32)
old_lp = (-np.log(A) + 0.2 * rng.normal(size=N)).astype(np.float32)
adv = rng.normal(size=N).astype(np.float32)
tgt = rng.normal(size=N).astype(np.float32)
d = [L.DevBuf(rctx, x) for x in (P, obs, idx, actions, old_lp, adv, tgt)]
dS = L.DevBuf(rctx, nbytes=32, dtype=np.float32)
dG = L.DevBuf(rctx, nbytes=P.size * 4, dtype=np.float32)
L._chk(rctx.lib.cbm_ppo_loss_grad(rctx.h, L._p(d[0].ptr), L._p(d[1].ptr), L._p(d[2].ptr), N, L._p(d[3].ptr), L._p(d[4].ptr), L._p(d[5].ptr),
                                  L._p(d[6].ptr), L._p(dS.ptr), L._p(dG.ptr), None, None))
logits, value, acts = oracle.resnet_forward(P, A, obs, idx=idx, save_acts=True)
stats, dlog, dval = oracle.ppo_loss_head(logits, value, actions, old_lp, adv, tgt)
grads_o = oracle.resnet_backward(P, A, obs, idx, acts, dlog, dval)
g = dG.download()
for name, (o, shp) in oracle.resnet_layout(A).items():
    n = int(np.prod(shp))
    ref = grads_o[o:o + n]
    err = np.abs(g[o:o + n] - ref)
    print(f"{name:28s} relerr {err.max() / max(np.abs(ref).max(), 1e-7):.2e}  nbad {(err > 1e-5 * np.abs(ref).max()).sum()}/{n}", flush=True)
    if "Conv_0.w" in name and name.startswith("seq0") and err.max() > 1e-4:
        e = (err.reshape(shp) > 1e-5 * np.abs(ref).max())
        print("   bad by tap:", e.reshape(9, -1).sum(1), " by ci:", e.reshape(9, shp[2], shp[3]).sum((0, 2)), " by co:", e.reshape(-1, shp[3]).sum(0))
