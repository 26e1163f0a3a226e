"""IMPALA (BASELINE configs[2] shape: a0-l0, E=120, T=128, Nature-CNN, optional bf16 forward): pipelined step time and rollout / update alone."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cleanba_amd.lib as L
import cleanba_amd.model as M
import cleanba_amd.prng as prng
E, T, A = 120, int(os.environ.get("T", "128")), 18
cfg = L.default_config(L.ALGO_IMPALA)
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions = E, 1, T, A
cfg.forward_bf16 = int(os.environ.get("BF16", "0"))
ctx = L.Context(cfg)
key = prng.prng_key(1)
key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_nature_params(A, nk, ak, ck))
ctx.actor_set_key(0, key)
ctx.actor_env_reset_device(0, 1)
lkey = key.copy()
lrs = np.full(4, 6e-4, np.float32)
first = [True]

def rollout():
    ctx.actor_begin_rollout(0, True); ctx.actor_rollout_device(0, T + 1 if first[0] else T); first[0] = False; ctx.actor_commit(0)

def update():
    global lkey
    ctx.learner_wait()
    lkey, _ = ctx.learner_update(lkey, lrs, lrs, lrs, want_stats=False)

rollout()
# warm up for >= 0.3 s of GPU work: on a fresh box the first process still loads code objects / ramps clocks after 2-3 short steps, which an
# 8-step sample at T = 20 (21 ms) then measures instead of the steady state (seen: 8.7 vs 2.67 ms per step)
for _ in range(max(2, 4000 // T)): rollout(); update()
ctx.sync(); t0 = time.perf_counter()
N = max(8, 1024 // T)
hr = hu = 0.0
for _ in range(N):
    a = time.perf_counter(); rollout(); b = time.perf_counter(); update(); hr += b - a; hu += time.perf_counter() - b
ctx.sync(); dt = (time.perf_counter() - t0) / N
print(f"IMPALA E={E} T={T} bf16={cfg.forward_bf16}: pipelined {dt*1e3:.2f} ms/step = {E*T/dt/1e3:.1f} k env-steps/s")
print(f"  host time inside the calls: rollout {hr/N*1e3:.2f} ms, update {hu/N*1e3:.2f} ms per step")
tr = tu = 0.0
for _ in range(4):
    ctx.sync(); t0 = time.perf_counter(); rollout(); ctx.sync(); tr += time.perf_counter() - t0
    t0 = time.perf_counter(); update(); ctx.sync(); tu += time.perf_counter() - t0
print(f"  rollout alone {tr/4*1e3:.2f} ms   update alone {tu/4*1e3:.2f} ms")
ctx.close()
