"""Where the envpool-API path's time per 120-env step goes (GPU box): host env step, obs copy, pageable vs registered H2D + inference + D2H,
python bookkeeping.  usage: python tools/host_path_probe.py"""
import ctypes as C
import os
import sys
import time

if os.environ.get("PROBE_NODE"):   # pin the whole process (and the pages it touches first) to one socket of the GPU box
    n = int(os.environ["PROBE_NODE"])
    os.sched_setaffinity(0, set(range(64 * n, 64 * n + 64)) | set(range(128 + 64 * n, 192 + 64 * n)))

import numpy as np

sys.path.insert(0, ".")
from cleanba_amd import lib as L  # noqa: E402
from cleanba_amd.args import parse_args  # noqa: E402
from cleanba_amd.trainer import make_config  # noqa: E402
from cleanba_amd.model import init_nature_params  # noqa: E402

E, T = 120, 128
args = parse_args(["--local-num-envs", str(E), "--num-steps", str(T), "--network", "nature"], "ppo")
cfg = make_config(args, "ppo")
ctx = L.Context(cfg)
ctx.set_params(init_nature_params(18, [0, 1], [0, 2], [0, 3]))
ctx.actor_set_key(0, np.array([0, 1], np.uint32))
st, obs = L.synth_env_reset_host(1, E)
actions = np.zeros(E, np.int32)
done = np.zeros(E, np.uint8)


def tm(f, n=100):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6


print("host env step      : %7.1f us" % tm(lambda: L.synth_env_step_host(1, st, obs, actions)))
print("obs.copy()         : %7.1f us" % tm(lambda: obs.copy()))


def steps(c, o):
    c.actor_begin_rollout(0, False)
    c.actor_step_host(0, o, done, None, None, actions)
    t0 = time.perf_counter()
    for _ in range(T - 1):
        c.actor_step_host(0, o, done, None, None, actions)
    return (time.perf_counter() - t0) / (T - 1) * 1e6


print("step_host pageable : %7.1f us" % steps(ctx, obs))
ctx2 = L.Context(cfg)
ctx2.set_params(init_nature_params(18, [0, 1], [0, 2], [0, 3]))
ctx2.actor_set_key(0, np.array([0, 1], np.uint32))
t0 = time.perf_counter()
ctx2.host_register(obs)
print("hipHostRegister    : %7.1f us (once)" % ((time.perf_counter() - t0) * 1e6))
print("step_host pinned   : %7.1f us" % steps(ctx2, obs))
ctx2.host_unregister(obs)
import os
print("cpu %d  affinity %d cpus" % (os.sched_getcpu() if hasattr(os, "sched_getcpu") else -1, len(os.sched_getaffinity(0))))
