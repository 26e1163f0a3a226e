#!/usr/bin/env python
"""Per-kernel timing of one learner minibatch (forward + loss + backward) at the headline minibatch size, using
the library's HIP-event profiler (cbm_profile_select): prints avg us and TFLOP/s per implicit-GEMM kernel.
`python tools/microbench.py [iters]`; under rocprofv3 pass `--plain` to skip the per-kernel passes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cleanba_amd.lib as L  # noqa: E402
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402
from bench import KERNELS  # noqa: E402

names_out = None
if "--names-out" in sys.argv:      # {id: "<kernel symbol> <functor type>"} of what the library launched (for tools/pmc_traffic.py)
    i = sys.argv.index("--names-out")
    names_out = sys.argv[i + 1]
    del sys.argv[i:i + 2]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
iters = int(args[0]) if args else 4
plain = "--plain" in sys.argv
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
ctx.learner_minibatch_grad(0, 0)
ctx.sync()
t0 = time.time()
for i in range(iters):
    ctx.learner_minibatch_grad(0, i % 4)
ctx.sync()
ms = (time.time() - t0) / iters * 1e3
if names_out:
    import json
    ctx.profile_select(-2)
    ctx.learner_minibatch_grad(0, 0)
    ctx.profile_read_all(12)
    ctx.profile_select(-1)
    json.dump({str(k): ctx.profile_kernel_name(k) for k in KERNELS}, open(names_out, "w"), indent=1)
print(f"minibatch fwd+loss+bwd: {ms:.3f} ms  ({sum(f for _, f in KERNELS.values()) / ms / 1e9:.1f} TFLOP/s over all GEMM flops)")
if not plain:
    tot = 0.0
    for kid, (name, flops) in KERNELS.items():
        ctx.profile_select(kid)
        for i in range(iters):
            ctx.learner_minibatch_grad(0, i % 4)
        t, n = ctx.profile_read()
        us = t / n * 1e3
        tot += us
        print(f"  {name:12s} {us:8.1f} us  {flops / us / 1e6:6.1f} TF  ({100 * flops / us / 1e6 / 157.3:4.1f}% of f32 MFMA peak)")
    ctx.profile_select(-1)
    print(f"  sum of GEMM kernels {tot:.1f} us")
