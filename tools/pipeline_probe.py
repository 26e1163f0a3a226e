"""Where does the PPO step time go: learner alone, actor alone, both pipelined (bench.py's loop)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cleanba_amd.lib as L
import cleanba_amd.model as M
import cleanba_amd.prng as prng
E, T, A, EPOCHS, NMB = 120, 128, 18, 4, 4
NET = os.environ.get("NET", "nature")
cfg = L.default_config(L.ALGO_PPO)
if NET != "nature":
    cfg.network = L.NET_IMPALA_RESNET
    cfg.actor_dense_ksplit = 11
if os.environ.get("KSPLIT"):   # K segments of the actor's dense launch (Nature: 7 / 14; the numerics spec ships 14)
    cfg.actor_dense_ksplit = int(os.environ["KSPLIT"])
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions = E, 1, T, A
ctx = L.Context(cfg)
key = prng.prng_key(1)
key, nk, ak, ck = prng.split(key, 4)
ctx.set_params(M.init_params(NET if NET == "nature" else "impala_resnet", A, nk, ak, ck))
ctx.actor_set_key(0, key)
ctx.actor_env_reset_device(0, 1)
lkey = key.copy()
n_opt = EPOCHS * NMB
lrs = np.full(n_opt, 2.5e-4, np.float32); bc = [M.adam_bias_corrections(i + 1) for i in range(n_opt)]
b1 = np.array([b[0] for b in bc], np.float32); b2 = np.array([b[1] for b in bc], np.float32)

def rollout():
    ctx.actor_begin_rollout(0, True); ctx.actor_rollout_device(0, T); ctx.actor_commit(0)

def update():
    global lkey
    ctx.learner_wait()
    lkey, _ = ctx.learner_update(lkey, lrs, b1, b2, want_stats=False)

# pipelined (bench loop)
rollout()
for _ in range(2): rollout(); update()
ctx.sync(); t0 = time.perf_counter()
N = 6
for _ in range(N): rollout(); update()
ctx.sync(); dt = (time.perf_counter() - t0) / N
print(f"pipelined      : {dt*1e3:.2f} ms/step")
# threaded like the real trainer: one host thread per actor slot, the learner on the main thread
import threading
NT = 8
def actor_loop(n):
    for _ in range(n): rollout()
ctx.sync()
th = threading.Thread(target=actor_loop, args=(NT + 1,)); th.start()
update()                       # consumes the rollout left over from the loop above
ctx.sync(); t0 = time.perf_counter()
for _ in range(NT): update()
ctx.sync(); dt = (time.perf_counter() - t0) / NT
th.join()
print(f"threaded       : {dt*1e3:.2f} ms/step")
update(); ctx.sync()
# serialized: rollout, sync, update, sync
tr = tu = 0.0
for _ in range(N):
    ctx.sync(); t0 = time.perf_counter(); rollout(); ctx.sync(); tr += time.perf_counter() - t0
    t0 = time.perf_counter(); update(); ctx.sync(); tu += time.perf_counter() - t0
print(f"rollout alone  : {tr/N*1e3:.2f} ms   update alone: {tu/N*1e3:.2f} ms   sum {(tr+tu)/N*1e3:.2f}")
# host enqueue cost (no sync inside)
ctx.sync(); t0 = time.perf_counter(); rollout(); te = time.perf_counter() - t0; ctx.sync()
update(); ctx.sync()
print(f"host time to enqueue one rollout: {te*1e3:.2f} ms")
ctx.close()
