"""The actor step in isolation: `iters` 128-step rollouts of 120 envs with nothing else on the GPU (device env), ms per rollout and us per env-step
batch; under `rocprofv3 --kernel-trace --stats` the per-kernel averages of the five launches of a step.  ALGO=impala, NET=resnet select the others.
usage: python tools/actor_probe.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cleanba_amd.lib as L  # noqa: E402
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
E, T, A = 120, 128, 18
NET, ALGO = os.environ.get("NET", "nature"), os.environ.get("ALGO", "ppo")
cfg = L.default_config(L.ALGO_PPO if ALGO == "ppo" else L.ALGO_IMPALA)
if NET != "nature":
    cfg.network, cfg.actor_dense_ksplit = L.NET_IMPALA_RESNET, 11
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions, cfg.ring_depth = E, 1, T, A, 2
ctx = L.Context(cfg)
key = prng.prng_key(1)
key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_params(NET if NET == "nature" else "impala_resnet", A, nk, ak, ck))
ctx.actor_set_key(0, key)
ctx.actor_env_reset_device(0, 1)
import ctypes as C  # noqa: E402


def rollout(n):
    # the actor alone: ring entries are recycled without a learner by marking the update done (cbm_params_publish_external bumps updates_done)
    ctx.actor_begin_rollout(0, True)
    ctx.actor_rollout_device(0, n)
    ctx.actor_commit(0)
    ctx.params_publish_external(ctx.buffer("params")[0])


rollout(T + (1 if ALGO != "ppo" else 0))
for _ in range(2):
    rollout(T)
ctx.sync()
t0 = time.perf_counter()
for _ in range(iters):
    rollout(T)
ctx.sync()
dt = (time.perf_counter() - t0) / iters
print(f"{ALGO} {NET}: rollout alone {dt * 1e3:.3f} ms = {dt / T * 1e6:.2f} us per 120-env step ({iters} rollouts)")
ctx.close()
