#!/usr/bin/env python
"""One IMPALA-ResNet learner minibatch (3840 frames: forward + fused heads/loss + backward) in isolation: ms per minibatch and the TFLOP/s
over the executed flops.  `python tools/rn_microbench.py [iters]`; run under rocprofv3 --kernel-trace for the per-kernel split
(tools/rocprof_summary.py).  CBM_SO selects a variant build (tools/variants.sh)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cleanba_amd.lib as L  # noqa: E402
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
E, T, A = 120, 128, 18
cfg = L.default_config(L.ALGO_PPO)
cfg.network = L.NET_IMPALA_RESNET
cfg.actor_dense_ksplit = 11
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = E, 1, T
ctx = L.Context(cfg)
key = prng.prng_key(1)
key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_params("impala_resnet", A, nk, ak, ck))
ctx.actor_set_key(0, key)
ctx.actor_env_reset_device(0, 1)
ctx.actor_begin_rollout(0, False)
ctx.actor_rollout_device(0, T)
ctx.actor_commit(0)
ctx.learner_wait()
k = ctx.learner_prepare(key)
k = ctx.learner_epoch_begin(k)
ctx.learner_minibatch_grad(0, 0)
ctx.sync()
t0 = time.time()
for i in range(iters):
    ctx.learner_minibatch_grad(0, i % 4)
ctx.sync()
ms = (time.time() - t0) / iters * 1e3
# executed flops per frame: forward 15 convs + dense + heads; backward 2x minus the first conv's input gradient
conv = lambda h, ci, co: 2.0 * h * h * 9 * ci * co  # noqa: E731
fwd = conv(84, 4, 16) + 4 * conv(42, 16, 16) + conv(42, 16, 32) + 4 * conv(21, 32, 32) + conv(21, 32, 32) + 4 * conv(11, 32, 32) + 2.0 * 3872 * 256 + 2.0 * 256 * (A + 1)
flops = (3 * fwd - conv(84, 4, 16)) * E * T / 4
print(f"{os.path.basename(L.SO_PATH):28s} resnet minibatch fwd+loss+bwd: {ms:7.3f} ms   {flops / ms / 1e9:6.1f} TFLOP/s ({flops / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak)")
