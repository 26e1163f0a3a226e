#!/usr/bin/env python
"""conv1 at learner size: the exact-product kernels (uint8 x three-term bf16 on the bf16 matrix cores, cbm_config.conv1_fp32_chain = 0) against the
fp32 fmaf chain (= 1, bit-identical to the oracle) on one 3840-frame PPO minibatch: logits / values / loss statistics / every gradient tensor.
GPU box: python tools/conv1_exact_ab.py [frames]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cleanba_amd.lib as L  # noqa: E402
from helpers import make_frames, make_params  # noqa: E402

A, E, T = 18, 120, 128
MB = int(sys.argv[1]) if len(sys.argv) > 1 else E * T // 4


def run(chain):
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, 1, T
    cfg.conv1_fp32_chain = chain
    ctx = L.Context(cfg)
    try:
        rng = np.random.default_rng(101)
        pool = make_frames(MB, 102)
        P = make_params(A, 103)
        idx = rng.permutation(MB).astype(np.int32)
        actions = rng.integers(0, A, MB).astype(np.int32)
        old_lp = (-np.log(A) + 0.2 * rng.normal(size=MB)).astype(np.float32)
        adv = rng.normal(size=MB).astype(np.float32)
        tgt = rng.normal(size=MB).astype(np.float32)
        d = [L.DevBuf(ctx, x) for x in (P, pool, idx, actions, old_lp, adv, tgt)]
        dS = L.DevBuf(ctx, nbytes=32, dtype=np.float32)
        dG = L.DevBuf(ctx, nbytes=P.size * 4, dtype=np.float32)
        dLg = L.DevBuf(ctx, nbytes=MB * A * 4, dtype=np.float32, shape=(MB, A))
        dV = L.DevBuf(ctx, nbytes=MB * 4, dtype=np.float32)
        L._chk(ctx.lib.cbm_ppo_loss_grad(ctx.h, L._p(d[0].ptr), L._p(d[1].ptr), L._p(d[2].ptr), MB, L._p(d[3].ptr), L._p(d[4].ptr),
                                         L._p(d[5].ptr), L._p(d[6].ptr), L._p(dS.ptr), L._p(dG.ptr), L._p(dLg.ptr), L._p(dV.ptr)))
        return dLg.download().copy(), dV.download().copy(), dS.download()[:5].copy(), dG.download().copy()
    finally:
        ctx.close()


os.environ.pop("CBM_CONV1_EXACT", None)
import cleanba_amd.model as M  # noqa: E402


def report(tag, ref, got):
    (lg0, v0, s0, g0), (lg1, v1, s1, g1) = ref, got
    print("==", tag)
    print("logits: max|ref| %.3e  max|d| %.3e   values: %.3e   stats: %.3e" % (np.abs(lg0).max(), np.abs(lg1 - lg0).max(), np.abs(v1 - v0).max(), np.abs(s1 - s0).max()))
    for name, (o, shp) in M.nature_layout(A)[0].items():
        n = int(np.prod(shp))
        print("  grad %-10s max|ref| %.3e  max|d| / max|ref| %.3e" % (name, np.abs(g0[o:o + n]).max(), np.abs(g1[o:o + n] - g0[o:o + n]).max() / max(np.abs(g0[o:o + n]).max(), 1e-30)))


print("frames", MB)
r3, r0, r1, r2 = run(3), run(0), run(1), run(2)
report("exact weight gradient behind the chain forward (same ReLU masks) vs all-chain", r3, r1)
report("exact forward + chain weight gradient vs all-exact (same forward)", r0, r2)
report("all-exact vs all-chain (ReLU masks of pre-activations within ~1e-7 of zero flip)", r3, r0)
